// Implicit-GEMM convolution on the CDNA4 matrix cores (wave64 MFMA 32x32).
//
//   out[m][n] = epi( sum_k A(m,k) * W[n][k] ),  m = output row (channels-last), n = output channel.
//
// One 256-thread workgroup (4 waves) owns a BM x BN output tile; the K loop walks 32-wide
// slices of the (tap, channel) axis.  Because activations are channels-last, a K slice of one
// tap is a contiguous 128-byte (fp32) / 64-byte (bf16) run of one source row, so the A tile is
// gathered with 16-byte loads, transformed in registers (GroupNorm+SiLU prologue: one fma and
// one SiLU per element from a per-(clip,channel) table in LDS) and staged in LDS; the W tile is
// K-contiguous too.  Global loads of slice t+1 are issued before the MFMAs of slice t and
// written to LDS after them (issue-early / write-late).
//
//   fp32 path : v_mfma_f32_32x32x2_f32  (exact fp32 fma chain; K order permuted {s, 16+s})
//   bf16 path : v_mfma_f32_32x32x16_bf16 (fp32 accumulate)
#include <cstring>
#include <set>
#include <string>

#include "common.h"
#include "kernels.h"

namespace sf {

namespace {

constexpr int BK = 32;

template <typename T> struct Lds {
  // row stride of an LDS tile in elements: 32 + one 16-byte pad
  static constexpr int LD = BK + 16 / (int)sizeof(T);
};

struct RowState {  // per staged A row held by a thread
  int valid_m;     // m < M
  int b_rel;       // clip index relative to the tile's first clip (GN table)
  int p0;          // 1-D: l*stride - pad ;  video: unused
  int base;        // 1-D: b*Lsrc ; video: n
  int t, h, w;     // video coordinates (already multiplied by stride, minus pad)
};

template <typename T, int BM, int BN, int WM_, int WN_, bool SCALAR_A>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmArgs a) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int VPR = BK / VEC;        // 16-byte vectors per tile row
  constexpr int RPP = 256 / VPR;       // rows staged per pass
  constexpr int PA = BM / RPP;
  constexpr int PB = (BN + RPP - 1) / RPP;   // BN < RPP: only the first BN*VPR threads stage W
  static_assert(PA >= 1 && BM % RPP == 0, "tile too small for the staging pattern");
  constexpr int LD = Lds<T>::LD;
  constexpr int WTM = BM / WM_, WTN = BN / WN_;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr bool FAST = sizeof(T) == 2;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *As = reinterpret_cast<T *>(smem);
  T *Bs = As + BM * LD;
  float2 *tab = reinterpret_cast<float2 *>(Bs + BN * LD);  // GN: (scale, shift) per (clip, channel)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave / WN_, wc = wave % WN_;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  const T *src = static_cast<const T *>(a.src);
  const T *src2 = static_cast<const T *>(a.src2);
  const T *wgt = static_cast<const T *>(a.w);

  const int srow = tid / VPR;   // row within a pass
  const int svec = tid % VPR;   // vector within the row

  // ---- per-row state of the A rows this thread stages -------------------------------------
  RowState rs[PA];
  const int b_first = (a.geom == 0) ? (m0 / a.Lout) : 0;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int m = m0 + i * RPP + srow;
    rs[i].valid_m = m < a.M;
    int mm = rs[i].valid_m ? m : 0;
    if (a.geom == 0) {
      int b = mm / a.Lout;
      int l = mm - b * a.Lout;
      rs[i].b_rel = b - b_first;
      rs[i].p0 = l * a.stride - a.pad;
      rs[i].base = b * a.Lsrc;
      rs[i].t = rs[i].h = rs[i].w = 0;
    } else {
      int w_ = mm % a.Wo;
      int r = mm / a.Wo;
      int h_ = r % a.Ho;
      r /= a.Ho;
      int t_ = r % a.To;
      int n_ = r / a.To;
      rs[i].b_rel = 0;
      rs[i].p0 = 0;
      rs[i].base = n_;
      rs[i].t = t_ * a.st - a.pt;
      rs[i].h = h_ * a.sh - a.ph;
      rs[i].w = w_ * a.sw - a.pw;
    }
  }

  // ---- GroupNorm+SiLU prologue table ----------------------------------------------------------
  if (a.pro == 1) {
    const int b_last = min(a.M - 1, m0 + BM - 1) / a.Lout;
    const int nb = b_last - b_first + 1;
    float2 *mr = tab + (size_t)a.cin * nb;  // (mean, rstd) per (clip, group), behind the table
    const int cpg = a.cin / a.G;
    for (int idx = tid >> 5; idx < nb * a.G; idx += 8) {   // one half-wave per (clip, group)
      const int bl = idx / a.G, g = idx - bl * a.G;
      const float *sl = a.stats + ((size_t)(b_first + bl) * a.nch) * a.G * 2 + g * 2;
      const float2 r = gn_merge32(sl, a.G, a.nch, a.chunk_rows, a.Lsrc, cpg, a.eps, tid & 31);
      if ((tid & 31) == 0) mr[idx] = r;
    }
    __syncthreads();
    for (int idx = tid; idx < nb * a.cin; idx += 256) {
      int bl = idx / a.cin, c = idx - bl * a.cin;
      float2 s = mr[bl * a.G + c / cpg];
      float sc = s.y * a.gamma[c];
      tab[idx] = make_float2(sc, a.beta[c] - s.x * sc);
    }
    __syncthreads();
  }

  // ---- accumulators ---------------------------------------------------------------------------
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Vec16<T> ra[PA], rb[PB];
  int bvalid[PB];
  int rvalid[PA];  // source row in range (bit) for the prefetched slice
  int rci0 = 0;    // first channel of the prefetched slice (GN table index)
  int rsecond = 0; // slice comes from src2

  const int nkt = (a.K + BK - 1) / BK;
  const int k_taps = a.taps * a.cin;

  auto prefetch = [&](int kt) {
    const int k0 = kt * BK;
    // ---- W tile ----
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int n = n0 + i * RPP + srow;
      int k = k0 + svec * VEC;
      if (i * RPP + srow >= BN) n = a.N;  // outside the tile
      // unconditional load from a clamped address (K is a multiple of 8 on every path); zeroed at stage time
      bvalid[i] = (n < a.N) && (k + VEC <= a.K);
      rb[i] = ld16<T>(wgt + (size_t)min(n, a.N - 1) * a.K + min(k, a.K - VEC));
    }
    // ---- A tile ----
    if constexpr (!SCALAR_A) {
      if (k0 < k_taps) {
        const int tap = k0 / a.cin;
        const int ci0 = k0 - tap * a.cin;
        rci0 = ci0 + svec * VEC;
        rsecond = 0;
        int dt = 0, dh = 0, dw = 0;
        if (a.geom == 1) {
          dw = tap % a.kw;
          int r = tap / a.kw;
          dh = r % a.kh;
          dt = r / a.kh;
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          int ok = rs[i].valid_m;
          size_t row;
          if (a.geom == 0) {
            int p = rs[i].p0 + tap;
            const int pmax = (a.Lsrc << a.up_shift) - 1;
            ok = ok && p >= 0 && p <= pmax;
            row = (size_t)(rs[i].base + (min(max(p, 0), pmax) >> a.up_shift));
          } else {
            int ti = rs[i].t + dt, hi = rs[i].h + dh, wi = rs[i].w + dw;
            ok = ok && ti >= 0 && ti < a.Ti && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
            row = ((size_t)(rs[i].base * a.Ti + min(max(ti, 0), a.Ti - 1)) * a.Hi + min(max(hi, 0), a.Hi - 1)) * a.Wi + min(max(wi, 0), a.Wi - 1);
          }
          rvalid[i] = ok;
          ra[i] = ld16<T>(src + row * a.src_ld + rci0);  // unconditional (clamped row); zeroed at stage time
        }
      } else {
        const int ci0 = k0 - k_taps + svec * VEC;
        rsecond = 1;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const int m = min(m0 + i * RPP + srow, a.M - 1);
          rvalid[i] = rs[i].valid_m;
          ra[i] = ld16<T>(src2 + (size_t)m * a.src2_ld + ci0);
        }
      }
    } else {
      // per-element (tap, channel) decode: thin-channel layers such as the RGB stem
      rsecond = 0;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        Vec16<T> v = zero16<T>();
        rvalid[i] = 0;
        if (rs[i].valid_m) {
          for (int j = 0; j < VEC; ++j) {
            int k = k0 + svec * VEC + j;
            if (k >= k_taps) break;
            int tap = k / a.cin, ci = k - tap * a.cin;
            int ok;
            size_t row;
            if (a.geom == 0) {
              int p = rs[i].p0 + tap;
              ok = p >= 0 && p < (a.Lsrc << a.up_shift);
              row = (size_t)(rs[i].base + (max(p, 0) >> a.up_shift));
            } else {
              int dw = tap % a.kw;
              int r = tap / a.kw;
              int dh = r % a.kh;
              int dt = r / a.kh;
              int ti = rs[i].t + dt, hi = rs[i].h + dh, wi = rs[i].w + dw;
              ok = ti >= 0 && ti < a.Ti && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
              row = ((size_t)(rs[i].base * a.Ti + max(ti, 0)) * a.Hi + max(hi, 0)) * a.Wi + max(wi, 0);
            }
            if (ok) v.set(j, to_f(src[row * a.src_ld + ci]));
          }
        }
        ra[i] = v;
      }
    }
  };

  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (i * RPP + srow < BN) st16<T>(Bs + (i * RPP + srow) * LD + svec * VEC, bvalid[i] ? rb[i] : zero16<T>());
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      Vec16<T> v = (SCALAR_A || rvalid[i]) ? ra[i] : zero16<T>();
      if (a.pro == 1 && !rsecond && rvalid[i]) {
        const float2 *tb = tab + (size_t)rs[i].b_rel * a.cin + rci0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float2 sd = tb[j];
          float y = fmaf(v.get(j), sd.x, sd.y);
          v.set(j, silu_t<FAST>(y));
        }
      }
      st16<T>(As + (i * RPP + srow) * LD + svec * VEC, v);
    }
  };

  const int fr = lane & 31, fh = lane >> 5;

  auto compute = [&]() {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        using frag = typename Frag16<T>::type;
        frag af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const frag *>(As + (wr * WTM + i * 32 + fr) * LD + 16 * s + 8 * fh);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = *reinterpret_cast<const frag *>(Bs + (wc * WTN + j * 32 + fr) * LD + 16 * s + 8 * fh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mfma32x16(af[i], bfr[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const f32x4 *>(As + (wr * WTM + i * 32 + fr) * LD + 16 * fh + 4 * q);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = *reinterpret_cast<const f32x4 *>(Bs + (wc * WTN + j * 32 + fr) * LD + 16 * fh + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop ------------------------------------------------------------------------------
  prefetch(0);
  stage();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) prefetch(kt + 1);
    compute();
    __syncthreads();
    if (more) {
      stage();
      __syncthreads();
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  // Every operand load is UNCONDITIONAL (indices clamped into range) and batched per 32x32 tile, so the 16
  // residual / per-clip loads of a lane are all in flight together; only the stores are predicated.
  T *out = static_cast<T *>(a.out);
  const T *res = static_cast<const T *>(a.res);
  const bool has_res = res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr, has_b = has_bs || has_ba;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wc * WTN + j * 32 + fr;
    const int nc = min(n, a.N - 1);
    const bool real = n < a.N;
    const float bias = a.bias ? a.bias[nc] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float rv[16], sv[16], av[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh, a.M - 1);
        rv[r] = has_res ? to_f(res[(size_t)m * a.res_ld + nc]) : 0.f;
        const int b = has_b ? m / a.Lout : 0;
        sv[r] = has_bs ? a.bscale[(size_t)b * a.bscale_ld + nc] : 1.f;
        av[r] = has_ba ? a.badd[(size_t)b * a.badd_ld + nc] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        float v = (acc[i][j][r] + bias) * sv[r] + rv[r] + av[r];
        v = real ? apply_act(v, a.act) : 0.f;
        if (m < a.M && n < a.n_store) {
          if (a.out_f32) static_cast<float *>(a.out)[(size_t)m * a.out_ld + n] = v;
          else out[(size_t)m * a.out_ld + n] = from_f<T>(v);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int WM_, int WN_, bool SC>
hipError_t launch_cfg(const ConvGemmArgs &a, hipStream_t s) {
  constexpr int LD = Lds<T>::LD;
  size_t lds = (size_t)(BM + BN) * LD * sizeof(T);
  if (a.pro == 1) {
    int nb = min(a.M / a.Lout + (a.M % a.Lout ? 1 : 0), BM / a.Lout + 2);
    lds += (size_t)nb * (a.cin + a.G) * sizeof(float2);
  }
  dim3 grid((a.M + BM - 1) / BM, (a.n_store + BN - 1) / BN);
  auto kern = conv_gemm_kernel<T, BM, BN, WM_, WN_, SC>;
  static bool big_lds_enabled = false;  // one-time opt-in to > 48 KiB of dynamic LDS (never inside graph capture)
  if (!big_lds_enabled) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return e;
    big_lds_enabled = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

// tile variants: 0 scalar-A 64x64, 1 128x32, 2 128x64, 3 64x64, 4 128x128
int pick_variant(const ConvGemmArgs &a);
int pick_variant(const ConvGemmArgs &a) {
  if (g_conv_gemm_force.path == 1 && g_conv_gemm_force.tile >= 1 && g_conv_gemm_force.tile <= 4) return g_conv_gemm_force.tile;
  const bool scalar_a = (a.cin % BK) != 0 || (a.cin2 % BK) != 0;
  if (scalar_a) return (a.cin2 != 0 || a.pro != 0) ? -1 : 0;
  const long M = a.M, N = a.n_store;
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  if (N <= 32) return 1;
  if (N <= 64) return blocks(128, 64) >= 512 ? 2 : 3;
  if (blocks(128, 128) >= 512) return 4;
  if (blocks(128, 64) >= 384) return 2;
  return 3;
}

template <typename T> hipError_t dispatch(const ConvGemmArgs &a, hipStream_t s) {
  switch (pick_variant(a)) {
    case 0: return launch_cfg<T, 64, 64, 2, 2, true>(a, s);
    case 1: return launch_cfg<T, 128, 32, 4, 1, false>(a, s);
    case 2: return launch_cfg<T, 128, 64, 2, 2, false>(a, s);
    case 3: return launch_cfg<T, 64, 64, 2, 2, false>(a, s);
    case 4: return launch_cfg<T, 128, 128, 2, 2, false>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool conv_gemm_supported(int dt, const ConvGemmArgs &a) {
  (void)dt;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
  if (a.pro == 1) {
    if (a.geom != 0 || a.cin % BK || a.cin % a.G) return false;
    int nb = min(a.M / a.Lout + 1, 128 / a.Lout + 2);
    if ((size_t)nb * (a.cin + a.G) * 8 > 96 * 1024) return false;
  }
  return true;
}

// short activations (the classic tiling would leave most CUs idle) go to the wave-split-K kernel
hipError_t launch_conv_gemm_sk(int dt, const ConvGemmArgs &a, hipStream_t s);
int conv_gemm_sk_variant(const ConvGemmArgs &a);
bool conv_gemm_fast_ok(int dt, const ConvGemmArgs &a);
bool conv_gemm_wp_ok(int dt, const ConvGemmArgs &a);
bool conv_gemm_prefers_wp(const ConvGemmArgs &a);

static bool use_sk(const ConvGemmArgs &a) {
  const int v = pick_variant(a);
  if (v <= 0 || (a.K % 32)) return false;
  static const int bm[5] = {64, 128, 128, 64, 128}, bn[5] = {64, 32, 64, 64, 128};
  const long blocks = (long)((a.M + bm[v] - 1) / bm[v]) * ((a.n_store + bn[v] - 1) / bn[v]);
  return blocks < 256 && a.K >= 256;
}

// profiling label of a 16-bit kernel in the f16 build: the bf16 label with the type renamed (interned: labels are kept by pointer)
const char *label_for_dtype(int dt, const char *bf16_label) {
  if (dt != F16) return bf16_label;
  static std::set<std::string> pool;
  std::string s(bf16_label);
  const size_t at = s.find("bf16");
  if (at != std::string::npos) s.replace(at, 4, "f16");
  return pool.insert(s).first->c_str();
}

static const char *variant_name_bf16(int dt, const ConvGemmArgs &a);
// fp32 launches that carry split-fp16 weights: the families that honour them are labelled <x3,...> (the others multiply in fp32)
static const char *label_x3(const ConvGemmArgs &a, const char *f32_label) {
  if (!a.wx) return f32_label;
  std::string s(f32_label);
  if (s.rfind("conv_gemm_wp<f32", 0) != 0 && s.rfind("conv_gemm_fast<f32", 0) != 0) return f32_label;
  static std::set<std::string> pool;
  s.replace(s.find("f32"), 3, "x3");
  return pool.insert(s).first->c_str();
}
const char *conv_gemm_variant_name(int dt, const ConvGemmArgs &a) {
  return dt == F32 ? label_x3(a, variant_name_bf16(dt, a)) : label_for_dtype(dt, variant_name_bf16(dt, a));
}
// With a split image at hand (ConvGemmArgs::wx), does launch_conv_gemm(dt, a) take a kernel family that reads ONLY that image -- the
// macro tiles, the wave-private / lean 32x32 kernels and the register-staged kernel in split mode, exactly the families label_x3() and
// variant_name_bf16() label "<x3" -- so that ConvGemmArgs::w need not exist?  (The training step then packs no fp32 image and passes a
// null `w`: a wrong answer here is a memory fault at address 0, not a wrong number.)
bool conv_gemm_reads_split_only(int dt, const ConvGemmArgs &a_in) {
  ConvGemmArgs a = a_in;
  a.w = a.wx = reinterpret_cast<const void *>(16);   // probe: the decision depends on shapes and flags only
  return std::strstr(conv_gemm_variant_name(dt, a), "<x3") != nullptr;
}

// Macro tiles pay from ~80 tiles of 256x128 per launch: a launch then occupies ~1/3 of the CUs at 4+ TFLOP/s each, and the
// engine's second clip-parallel branch fills most of the rest (measured on BASELINE configs[2]: threshold 160 -> 138, 80 -> 145.5
// steps/s; alone on the chip the 64x64 kernel still wins below ~160 tiles, tools/gemm_mt.py).  K >= 256: the three-slot ring
// needs a few steps to reach steady state.
static long short_act_tiles() {   // tuning hook: below this many 64x64 tiles a GEMM goes to the wave-split-K / wave-private kernels
  static const long v = [] {
    const char *e = tune_env("SF_SHORT_TILES");
    const long t = e ? atol(e) : 0;
    return t > 0 ? t : 500L;
  }();
  return v;
}

bool conv_gemm_prefers_mt(const ConvGemmArgs &a) {
  static const int mode = [] {   // SF_MT=0 disables the kernel, SF_MT=2 prefers it wherever it is eligible (tuning / tests)
    const char *e = tune_env("SF_MT");
    return e ? atoi(e) : 1;
  }();
  if (mode == 0) return false;
  if (mode == 2) return true;
  static const int min_tiles = [] {   // tuning hook
    const char *e = tune_env("SF_MT_TILES");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 40;   // re-measured with the later tile variants: batch 32 without guidance 218 (80) -> 230 (40) steps/s, batch 16 335 -> 341,
                             // batch 32 with guidance and batch 8 unchanged
  }();
  // outputs of <= 64 columns: half-empty 128-wide tiles lose to the 64x64 kernel (346 vs 259 TFLOP/s on the onset net's 192 -> 64
  // temporal convolution); the 128x64 tile with two workgroups per CU (video geometry) wins (542 vs 770 us on that shape)
  if (a.n_store <= 64 && a.geom != 1) return false;
  const long tiles = (long)((a.M + 127) / 128) * ((a.n_store + 127) / 128);   // the 128x128 variant takes over below 160 tiles of 256x128
  // (shortest 1-D reduction: 256; 192 with the context padded to 64 columns was measured twice at -0.9 ... +1.6 % by workload and removed)
  return tiles >= 2 * min_tiles && a.K >= (a.geom == 1 ? 128 : 256);   // two workgroups per CU cover the short pipelines of the video geometry
}

static const char *variant_name_bf16(int dt, const ConvGemmArgs &a) {
  if (g_conv_gemm_force.path == 6 || (g_conv_gemm_force.path == 0 && conv_gemm_mt_wanted(dt, a))) return dt == F32 ? (a.wx ? "conv_gemm_mt<x3>" : "conv_gemm_mt<f32>") : conv_gemm_mt_name(a);
  static const char *names[2][5] = {{"conv_gemm<f32,64x64,scalarA>", "conv_gemm<f32,128x32>", "conv_gemm<f32,128x64>", "conv_gemm<f32,64x64>", "conv_gemm<f32,128x128>"},
                                    {"conv_gemm<bf16,64x64,scalarA>", "conv_gemm<bf16,128x32>", "conv_gemm<bf16,128x64>", "conv_gemm<bf16,64x64>", "conv_gemm<bf16,128x128>"}};
  static const char *sk_names[2][3] = {{"conv_gemm_sk<f32,64x64>", "conv_gemm_sk<f32,64x32>", "conv_gemm_sk<f32,32x32>"},
                                       {"conv_gemm_sk<bf16,64x64>", "conv_gemm_sk<bf16,64x32>", "conv_gemm_sk<bf16,32x32>"}};
  static const char *fast_names[2][3] = {{"conv_gemm_fast<f32,64x64>", "conv_gemm_fast<f32,64x32>", "conv_gemm_fast<f32,32x32>"},
                                         {"conv_gemm_fast<bf16,64x64>", "conv_gemm_fast<bf16,64x32>", "conv_gemm_fast<bf16,32x32>"}};
  const long t64 = (long)((a.M + 63) / 64) * ((a.n_store + 63) / 64);
  const bool short_act = t64 < short_act_tiles() && a.K >= 256 && (a.K % 32) == 0 && (a.cin % 32) == 0 && (a.cin2 % 32) == 0;
  if (!short_act) {
    V2Plan pl;
    if (conv_gemm_v2_plan(dt, a, pl)) return conv_gemm_v2_name(dt, pl);
  }
  static const char *wp_names[2][3] = {{"conv_gemm_wp<f32,32x32>", "conv_gemm_wp<f32,32x32>", "conv_gemm_wp<f32,32x32>"},
                                       {"conv_gemm_wp<bf16,64x64>", "conv_gemm_wp<bf16,64x32>", "conv_gemm_wp<bf16,32x32>"}};
  if ((short_act || use_sk(a)) && conv_gemm_prefers_wp(a) && conv_gemm_sk_variant(a) == 2 && g_conv_gemm_force.path == 0 && conv_gemm_rs_ok(dt, a))
    return dt == F32 ? "conv_gemm_rs<x3,32x32>" : "conv_gemm_rs<bf16,32x32>";
  if ((short_act || use_sk(a)) && conv_gemm_prefers_wp(a) && conv_gemm_wp_ok(dt, a)) return wp_names[dt == F32 ? 0 : 1][conv_gemm_sk_variant(a)];
  if (short_act || use_sk(a)) return (conv_gemm_fast_ok(dt, a) ? fast_names : sk_names)[dt == F32 ? 0 : 1][conv_gemm_sk_variant(a)];
  int v = pick_variant(a);
  return v < 0 ? "conv_gemm<invalid>" : names[dt == F32 ? 0 : 1][v];
}

ConvGemmForce g_conv_gemm_force;

// Row-LayerNorm fusion on the macro tiles (row partials in the epilogue, LayerNorm on the accumulator): parity-tested at op level, but in
// the two-branch step the launches it removes were hidden under the other branch's kernels and its epilogue work is not -- same-box A/B
// (profiles/r5_c_ab_mt_ln.txt): configs[2] 202 -> 203.5 steps/s, batch 32 without guidance 301 -> 291.  Taken only when the caller asks
// for it (ConvGemmArgs::mt_ln; the engine does not).

bool conv_gemm_emits_rowpart(int dt, const ConvGemmArgs &a) {
  if (g_conv_gemm_force.path != 0 || !conv_gemm_supported(dt, a)) return false;
  {   // long activations: the macro-tile kernel; its epilogue writes the row partials for column counts that are multiples of 32
    ConvGemmArgs plain = a;
    plain.rowpart_out = nullptr;
    if (conv_gemm_mt_wanted(dt, plain)) {
      const bool off = a.mt_ln == 0;
      ConvGemmArgs probe = a;   // (callers ask before they arm the launch)
      if (!probe.rowpart_out) {
        probe.rowpart_out = reinterpret_cast<float *>(16);
        probe.rowpart_nt = a.n_store / 32;
      }
      return !off && conv_gemm_mt_wanted(dt, probe);
    }
  }
  if ((a.n_store % 32) || a.n_store != a.N) return false;
  const long t64 = (long)((a.M + 63) / 64) * ((a.n_store + 63) / 64);
  const bool short_act = t64 < short_act_tiles() && a.K >= 256 && (a.K % 32) == 0 && (a.cin % 32) == 0 && (a.cin2 % 32) == 0;
  if (!(short_act || use_sk(a))) return false;                 // would go to v2 / the classic tiles
  if ((a.cin % 32) || (a.cin2 % 32) || (a.K % 32)) return false;
  if (conv_gemm_sk_variant(a) != 2) return false;              // 32x32 tiles only
  const bool prefer_wp = conv_gemm_prefers_wp(a);
  if (prefer_wp && conv_gemm_wp_ok(dt, a)) return true;        // wp 32x32 (its LDS footprint fits in both types)
  return conv_gemm_fast_ok(dt, a);
}

bool conv_gemm_emits_gnpart(int dt, const ConvGemmArgs &a) {
  ConvGemmArgs plain = a;
  plain.gnpart_out = nullptr;
  if (a.geom != 0 || a.Lout < 32 || !conv_gemm_emits_rowpart(dt, plain)) return false;
  return conv_gemm_prefers_wp(a) && conv_gemm_wp_ok(dt, a);   // the wave-private 32x32 kernel is the one that writes them
}

hipError_t launch_conv_gemm(int dt, const ConvGemmArgs &a, hipStream_t s) {
  if (!conv_gemm_supported(dt, a)) return hipErrorInvalidValue;
  if (a.src_x3 && (dt != F32 || g_conv_gemm_force.path != 0 || !conv_gemm_src_x3_ok(a))) return hipErrorInvalidValue;   // only the macro tiles read pre-split rows
  const ConvGemmForce &f = g_conv_gemm_force;
  if (f.path == 6) return launch_conv_gemm_mt(dt, a, s);
  if (f.path == 0 && conv_gemm_mt_wanted(dt, a)) return launch_conv_gemm_mt(dt, a, s);
  const long t64 = (long)((a.M + 63) / 64) * ((a.n_store + 63) / 64);
  const bool short_act = t64 < short_act_tiles() && a.K >= 256 && (a.K % 32) == 0 && (a.cin % 32) == 0 && (a.cin2 % 32) == 0;
  if (f.path == 4 || (f.path == 0 && !short_act)) {
    V2Plan pl;   // long activations: classic 2x2-wave tiles, channel counts that are multiples of 64, no prologue
    if (conv_gemm_v2_plan(dt, a, pl)) return launch_conv_gemm_v2(dt, a, pl, s);
    if (f.path == 4) return hipErrorInvalidValue;
  }
  if (f.path == 2 || f.path == 5 || (f.path == 0 && (short_act || use_sk(a)))) return launch_conv_gemm_sk(dt, a, s);
  return SF_DISPATCH_T(dt, dispatch<T>(a, s));
}

}  // namespace sf
