// Implicit-GEMM convolution for SHORT activations (few output rows, wide channels): the deep U-Net levels
// (M = B*L_d = 352..5632 rows, N = 128..1024, K up to 3072) and the 7x7 / 14x14 stages of the onset net.
//
// With so few rows a BM x BN grid of classic tiles cannot fill 256 CUs, and each tile walks a long K with
// one 32-wide slice in flight -- latency-bound (measured: 43 us for a 2.2 GFLOP layer).  Here the four waves
// of a workgroup SPLIT K instead of M/N: every iteration stages a 128-wide K chunk of A and W (16-32 KiB in
// flight per workgroup), wave w multiplies sub-slice [32w, 32w+32) into its own full BM x BN accumulator, and
// the four partial tiles are summed through LDS in the epilogue (deterministic order, no atomics, no second
// launch).  Small tiles (32x32 .. 64x64) give 200-700 workgroups on those layers.  W tiles are re-read by
// every row-tile, so the block->tile map keeps all row-tiles of one W panel on one XCD (shared L2).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int BKT = 128;  // K chunk per iteration (4 waves x 32)

struct RowSt {
  int valid_m, b_rel, p0, base, t, h, w;
};

// GEOM: 0 = 1-D rows, 1 = video rows; CAT: a second (concatenated) source follows the taps; PRO: GroupNorm+SiLU
// prologue.  Compile-time so that the gather in the K loop is straight-line code (no scalar branches around loads).
template <typename T, int BM, int BN, int GEOM, bool CAT, bool PRO>
__global__ __launch_bounds__(256) void conv_gemm_sk_kernel(const ConvGemmArgs a, const int mtiles, const int ntiles, const int swz) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int VPR = BKT / VEC;   // vectors per staged row (bf16 16, fp32 32)
  constexpr int RPP = 256 / VPR;   // rows per pass (16 / 8)
  constexpr int PA = BM / RPP, PB = BN / RPP;
  constexpr int LD = BKT + 16 / (int)sizeof(T);
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr bool FAST = sizeof(T) == 2;
  constexpr int LDR = BN + 4;      // fp32 row stride of the reduction buffer

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *As = reinterpret_cast<T *>(smem);
  T *Bs = As + BM * LD;
  constexpr size_t stage_bytes = (size_t)(BM + BN) * LD * sizeof(T);
  constexpr size_t red_bytes = (size_t)4 * BM * LDR * sizeof(float);
  constexpr size_t main_bytes = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  float2 *tab = reinterpret_cast<float2 *>(smem + main_bytes);
  float *red = reinterpret_cast<float *>(smem);  // reuses the staging area after the K loop

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- block -> (row tile, column tile); keep one W panel's row tiles on one XCD ---------------------
  int bid = blockIdx.x, mt, nt;
  if (swz) {  // ntiles % 8 == 0: XCD x (= bid % 8 under round-robin dispatch) owns column tiles == x (mod 8)
    const int xcd = bid & 7, j = bid >> 3;
    nt = xcd + 8 * (j / mtiles);
    mt = j % mtiles;
  } else {
    nt = bid / mtiles;
    mt = bid % mtiles;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const T *src = static_cast<const T *>(a.src);
  const T *src2 = static_cast<const T *>(a.src2);
  const T *wgt = static_cast<const T *>(a.w);
  const int srow = tid / VPR, svec = tid % VPR;

  RowSt rs[PA];
  const int b_first = (GEOM == 0) ? (m0 / a.Lout) : 0;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int m = m0 + i * RPP + srow;
    rs[i].valid_m = m < a.M;
    int mm = rs[i].valid_m ? m : 0;
    if constexpr (GEOM == 0) {
      int b = mm / a.Lout, l = mm - b * a.Lout;
      rs[i].b_rel = b - b_first;
      rs[i].p0 = l * a.stride - a.pad;
      rs[i].base = b * a.Lsrc;
      rs[i].t = rs[i].h = rs[i].w = 0;
    } else {
      int w_ = mm % a.Wo, r = mm / a.Wo;
      int h_ = r % a.Ho;
      r /= a.Ho;
      int t_ = r % a.To, n_ = r / a.To;
      rs[i].b_rel = 0;
      rs[i].p0 = 0;
      rs[i].base = n_;
      rs[i].t = t_ * a.st - a.pt;
      rs[i].h = h_ * a.sh - a.ph;
      rs[i].w = w_ * a.sw - a.pw;
    }
  }

  if constexpr (PRO) {  // GroupNorm+SiLU prologue table: (scale, shift) per (clip, channel)
    const int b_last = min(a.M - 1, m0 + BM - 1) / a.Lout;
    const int nb = b_last - b_first + 1;
    float2 *mr = tab + (size_t)a.cin * nb;
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // two register sets: chunk t+1 and t+2 are in flight while chunk t is multiplied (named sets, static indexing)
  struct RegSet {
    Vec16<T> ra[PA], rb[PB];
    int rvalid[PA], bvalid[PB];
    int rci, rsecond;
  };
  RegSet s0, s1;
  const int nkt = (a.K + BKT - 1) / BKT;
  const int k_taps = a.taps * a.cin;

  auto prefetch = [&](int kt, RegSet &R) {
    Vec16<T>(&ra)[PA] = R.ra;
    Vec16<T>(&rb)[PB] = R.rb;
    int(&rvalid)[PA] = R.rvalid;
    int(&bvalid)[PB] = R.bvalid;
    int &rci = R.rci;
    int &rsecond = R.rsecond;
    // Loads are UNCONDITIONAL from clamped (always mapped) addresses; out-of-range rows / columns are zeroed when
    // the registers are written to LDS.  A load under a per-lane branch would make hipcc wait for each one in turn.
    const int kv_raw = kt * BKT + svec * VEC;  // this thread's K position (a vector never straddles a tap: cin % 32 == 0)
    const bool k_ok = kv_raw < a.K;
    const int kv = k_ok ? kv_raw : a.K - VEC;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int n = n0 + i * RPP + srow;
      bvalid[i] = (n < a.N) && k_ok;
      rb[i] = ld16<T>(wgt + (size_t)min(n, a.N - 1) * a.K + kv);
    }
    if (!CAT || kv < k_taps) {
      const int tap = kv / a.cin;
      rci = kv - tap * a.cin;
      rsecond = 0;
      int dt = 0, dh = 0, dw = 0;
      if constexpr (GEOM == 1) {
        dw = tap % a.kw;
        int r = tap / a.kw;
        dh = r % a.kh;
        dt = r / a.kh;
      }
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        int ok = rs[i].valid_m;
        size_t row;
        if constexpr (GEOM == 0) {
          int p = rs[i].p0 + tap;
          const int pmax = (a.Lsrc << a.up_shift) - 1;
          ok = ok && p >= 0 && p <= pmax;
          row = (size_t)(rs[i].base + (min(max(p, 0), pmax) >> a.up_shift));
        } else {
          int ti = rs[i].t + dt, hi = rs[i].h + dh, wi = rs[i].w + dw;
          ok = ok && ti >= 0 && ti < a.Ti && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
          row = ((size_t)(rs[i].base * a.Ti + min(max(ti, 0), a.Ti - 1)) * a.Hi + min(max(hi, 0), a.Hi - 1)) * a.Wi + min(max(wi, 0), a.Wi - 1);
        }
        rvalid[i] = ok && k_ok;
        ra[i] = ld16<T>(src + row * a.src_ld + rci);
      }
    } else {
      rsecond = 1;
      const int ci2 = kv - k_taps;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int m = min(m0 + i * RPP + srow, a.M - 1);
        rvalid[i] = rs[i].valid_m && k_ok;
        ra[i] = ld16<T>(src2 + (size_t)m * a.src2_ld + ci2);
      }
    }
  };

  auto stage = [&](RegSet &R) {
    Vec16<T>(&ra)[PA] = R.ra;
    Vec16<T>(&rb)[PB] = R.rb;
    int(&rvalid)[PA] = R.rvalid;
    int(&bvalid)[PB] = R.bvalid;
    const int rci = R.rci, rsecond = R.rsecond;
#pragma unroll
    for (int i = 0; i < PB; ++i) st16<T>(Bs + (i * RPP + srow) * LD + svec * VEC, bvalid[i] ? rb[i] : zero16<T>());
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      Vec16<T> v = rvalid[i] ? ra[i] : zero16<T>();
      if (PRO && !rsecond && rvalid[i]) {
        const float2 *tb = tab + (size_t)rs[i].b_rel * a.cin + rci;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float2 sd = tb[j];
          v.set(j, silu_t<FAST>(fmaf(v.get(j), sd.x, sd.y)));
        }
      }
      st16<T>(As + (i * RPP + srow) * LD + svec * VEC, v);
    }
  };

  const int fr = lane & 31, fh = lane >> 5;
  const int kw0 = 32 * wave;  // this wave's sub-slice of the chunk

  auto compute = [&]() {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        using frag = typename Frag16<T>::type;
        frag af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const frag *>(As + (i * 32 + fr) * LD + kw0 + 16 * s + 8 * fh);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag *>(Bs + (j * 32 + fr) * LD + kw0 + 16 * s + 8 * fh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32x16(af[i], bfr[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(As + (i * 32 + fr) * LD + kw0 + 16 * fh + 4 * q);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const f32x4 *>(Bs + (j * 32 + fr) * LD + kw0 + 16 * fh + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  prefetch(0, s0);
  if (nkt > 1) prefetch(1, s1);
  for (int kt = 0; kt < nkt; kt += 2) {
    stage(s0);
    __syncthreads();
    if (kt + 2 < nkt) prefetch(kt + 2, s0);
    compute();
    __syncthreads();
    if (kt + 1 < nkt) {
      stage(s1);
      __syncthreads();
      if (kt + 3 < nkt) prefetch(kt + 3, s1);
      compute();
      __syncthreads();
    }
  }

  // ---- cross-wave K reduction through LDS, then a row-major epilogue -------------------------------
  float *myred = red + (size_t)wave * BM * LDR;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) myred[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + j * 32 + fr] = acc[i][j][r];
  __syncthreads();

  T *out = static_cast<T *>(a.out);
  const T *res = static_cast<const T *>(a.res);
  const bool has_res = res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
  constexpr int QN = BN / 4;
#pragma unroll
  for (int it = 0; it < (BM * QN + 255) / 256; ++it) {
    const int idx = tid + it * 256;
    const int ml = idx / QN, nq = idx - ml * QN;
    const int m = m0 + ml, nb = n0 + nq * 4;
    const bool live = idx < BM * QN && m < a.M && nb < a.n_store;
    const int mc = min(m, a.M - 1);
    // operands first (unconditional, clamped): all loads of this row segment are in flight together
    float bi[4], rv[4], sv[4], av[4];
    const int b = (has_bs || has_ba) ? mc / a.Lout : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int nc = min(nb + e, a.N - 1);
      bi[e] = a.bias ? a.bias[nc] : 0.f;
      rv[e] = has_res ? to_f(res[(size_t)mc * a.res_ld + nc]) : 0.f;
      sv[e] = has_bs ? a.bscale[(size_t)b * a.bscale_ld + nc] : 1.f;
      av[e] = has_ba ? a.badd[(size_t)b * a.badd_ld + nc] : 0.f;
    }
    const int mlc = min(ml, BM - 1);
    f32x4 v = *reinterpret_cast<const f32x4 *>(red + (size_t)mlc * LDR + nq * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      f32x4 t = *reinterpret_cast<const f32x4 *>(red + ((size_t)w * BM + mlc) * LDR + nq * 4);
      v[0] += t[0];
      v[1] += t[1];
      v[2] += t[2];
      v[3] += t[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nb + e;
      float x = (v[e] + bi[e]) * sv[e] + rv[e] + av[e];
      x = n < a.N ? apply_act(x, a.act) : 0.f;
      if (live && n < a.n_store) {
        if (a.out_f32) static_cast<float *>(a.out)[(size_t)m * a.out_ld + n] = x;
        else out[(size_t)m * a.out_ld + n] = from_f<T>(x);
      }
    }
  }
}

template <typename T, int BM, int BN, int GEOM, bool CAT, bool PRO> hipError_t launch_sk2(const ConvGemmArgs &a, hipStream_t s) {
  constexpr int LD = BKT + 16 / (int)sizeof(T);
  constexpr size_t stage_bytes = (size_t)(BM + BN) * LD * sizeof(T);
  constexpr size_t red_bytes = (size_t)4 * BM * (BN + 4) * sizeof(float);
  size_t lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  if (PRO) {
    int nb = min(a.M / a.Lout + (a.M % a.Lout ? 1 : 0), BM / a.Lout + 2);
    lds += (size_t)nb * (a.cin + a.G) * sizeof(float2);
  }
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.n_store + BN - 1) / BN;
  const int swz = (ntiles % 8 == 0) ? 1 : 0;
  auto kern = conv_gemm_sk_kernel<T, BM, BN, GEOM, CAT, PRO>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(256), lds, s, a, mtiles, ntiles, swz);
  return hipGetLastError();
}

template <typename T, int BM, int BN> hipError_t launch_sk(const ConvGemmArgs &a, hipStream_t s) {
  if (a.geom == 1) {
    if (a.cin2 != 0 || a.pro != 0) return hipErrorInvalidValue;
    return launch_sk2<T, BM, BN, 1, false, false>(a, s);
  }
  if (a.cin2 != 0) {
    if (a.pro != 0) return hipErrorInvalidValue;
    return launch_sk2<T, BM, BN, 0, true, false>(a, s);
  }
  if (a.pro == 1) return launch_sk2<T, BM, BN, 0, false, true>(a, s);
  return launch_sk2<T, BM, BN, 0, false, false>(a, s);
}

}  // namespace

// tile choice: the largest of 64x64 / 64x32 / 32x32 that still yields >= 224 workgroups
int conv_gemm_sk_variant(const ConvGemmArgs &a) {
  if ((g_conv_gemm_force.path == 2 || g_conv_gemm_force.path == 5) && g_conv_gemm_force.tile >= 0 && g_conv_gemm_force.tile <= 2)
    return g_conv_gemm_force.tile;
  // measured (tools/gemm_sweep.py): on short activations the smallest tile wins at every U-Net shape -- more
  // workgroups in flight matter more than operand reuse
  auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.n_store + bn - 1) / bn); };
  static const int exp_rule = [] {   // tuning hook (tools/): alternative tile rules under the two-branch bench
    const char *e = tune_env("SF_SK_TILE_RULE");
    return e ? atoi(e) : 0;
  }();
  if (exp_rule == 1 && blocks(32, 32) > 300 && a.K >= 512) return 1;
  if (exp_rule == 2 && a.K >= 1024) return 1;
  if (exp_rule == 3 && blocks(32, 32) > 300 && a.K >= 512) return 0;
  if (exp_rule == 4 && a.K >= 1024) return 0;
  if (blocks(32, 32) <= 4096) return 2;
  if (blocks(64, 32) <= 4096) return 1;
  return 0;
}

bool conv_gemm_fast_ok(int dt, const ConvGemmArgs &a);
hipError_t launch_conv_gemm_fast(int dt, const ConvGemmArgs &a, int variant, hipStream_t s);
bool conv_gemm_wp_ok(int dt, const ConvGemmArgs &a);
hipError_t launch_conv_gemm_wp(int dt, const ConvGemmArgs &a, int variant, hipStream_t s);

bool conv_gemm_prefers_wp(const ConvGemmArgs &a);
bool conv_gemm_rs_rows_ok(int64_t rows, int N) {
  ConvGemmArgs a;
  a.M = (int)rows;
  a.N = a.n_store = N;
  a.K = 1024;
  return conv_gemm_prefers_wp(a) && conv_gemm_sk_variant(a) == 2;
}

// Barrier-free wave-private pipelines (conv_gemm_wp.hip) win where few workgroups exist (cold weights, tools/gemm_cold.py):
// at <= 512 tiles of 32x32 -- every GEMM of depths 3-7 at batch 4 except the widest qkv projections -- the staged
// kernel's two barriers per chunk cost more than the operand sharing they buy.
bool conv_gemm_prefers_wp(const ConvGemmArgs &a) {
  static const long max_tiles = [] {   // tuning hook
    const char *e = tune_env("SF_WP_TILES");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 512L;
  }();
  const long tiles = (long)((a.M + 31) / 32) * ((a.n_store + 31) / 32);
  return (tiles <= max_tiles && a.K >= 256) || (a.M <= 512 && a.K >= 2048);
}

hipError_t launch_conv_gemm_sk(int dt, const ConvGemmArgs &a, hipStream_t s) {
  if ((a.cin % 32) || (a.cin2 % 32) || (a.K % 32)) return hipErrorInvalidValue;
  const int v = conv_gemm_sk_variant(a);
  // barrier-free wave-private pipelines where few tiles exist (conv_gemm_prefers_wp); with many tiles the staged kernel
  // wins because its loads are shared by more MFMA work per byte
  const bool prefer_wp = g_conv_gemm_force.path == 5 || (g_conv_gemm_force.path == 0 && conv_gemm_prefers_wp(a));
  // few 32x32 tiles and fragment-ordered weights at hand: the register-staged kernel (all loads of a wave up front)
  if (g_conv_gemm_force.path == 0 && prefer_wp && v == 2 && conv_gemm_rs_ok(dt, a)) return launch_conv_gemm_rs(dt, a, s);
  if (prefer_wp && conv_gemm_wp_ok(dt, a)) {
    hipError_t e = launch_conv_gemm_wp(dt, a, dt == F32 ? 2 : v, s);
    if (e != hipErrorInvalidValue) return e;
  }
  if (conv_gemm_fast_ok(dt, a)) return launch_conv_gemm_fast(dt, a, v, s);   // lean path (conv_gemm_fast.hip)
  if (dt == F32) {
    switch (v) {
      case 0: return launch_sk<float, 64, 64>(a, s);
      case 1: return launch_sk<float, 64, 32>(a, s);
      default: return launch_sk<float, 32, 32>(a, s);
    }
  }
  if (dt == F16) {
    switch (v) {
      case 0: return launch_sk<f16, 64, 64>(a, s);
      case 1: return launch_sk<f16, 64, 32>(a, s);
      default: return launch_sk<f16, 32, 32>(a, s);
    }
  }
  switch (v) {
    case 0: return launch_sk<bf16, 64, 64>(a, s);
    case 1: return launch_sk<bf16, 64, 32>(a, s);
    default: return launch_sk<bf16, 32, 32>(a, s);
  }
}

}  // namespace sf
