// Thin-level convolutions of the U-Net (C = 32 / 64 channels, long sequences: depths 1-2 of the reference
// config, main/module_diffusion.py -> a_unet ResnetItem / ModulationItem / InjectChannelsItem).
//
// At these depths a layer is 6-25 MB of activations against a few KB of weights: the bound is HBM/L2 streaming and
// the number of dependent memory round trips per workgroup, not MFMA.  The generic implicit-GEMM kernels re-apply
// GroupNorm+SiLU per tap and per column tile and pay several barriers for a 96-deep reduction.  Here instead:
//
//   * one workgroup = RW consecutive positions of one clip (all channels); the input rows (+ halo) are read ONCE,
//     the prologue is applied ONCE per element and the result is staged in LDS:
//         PRO 1  GroupNorm + SiLU   (statistics: per-chunk (mean, M2) partials of the producer, Chan-merged here)
//         PRO 2  LayerNorm over channels * (1 + scale) + shift     (the Modulation item; never materialised in HBM)
//   * the product is computed TRANSPOSED, D^T[c][r] = sum_k W[c][k] * act[r][k]: the weights are the MFMA A operand
//     (rows of the packed [N][K] matrix, 16 bytes per lane, kept in registers), the staged activations the B operand
//     (8 channels of one position per lane, one ds_read_b128 per tap), so every lane ends up holding 4 consecutive
//     channels x 4 of ONE position -> vector stores along the channel axis and lane-local GroupNorm partials;
//   * each wave owns a 32-position tile: no barrier inside the reduction;
//   * the epilogue adds bias / residual / per-clip bias and emits the (mean, M2) partial of its own output per
//     (workgroup, group), which is what the next GroupNorm consumes -- the separate statistics pass disappears.
//
// K is cut in slots of 8 channels: slot -> (tap, channel octet) for the convolution input, then the octets of the
// second source (InjectChannels concatenates [x, context]).  MFMA step s takes slot 2s on lanes 0-31 and slot 2s+1 on
// lanes 32-63 (bf16: one 32x32x16; fp32 parity path: eight 32x32x2).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

// waves per workgroup at most, by the channel count of the level (tuning hooks SF_THIN_WAVES32 / SF_THIN_WAVES64: 8 lets two workgroups
// of 512-position chunks share a CU -- the kernels hold ~100 registers, so a CU takes 16-20 waves of them)
static int thin_max_waves(int C) {
  static const int v32 = [] {
    const char *e = tune_env("SF_THIN_WAVES32");
    const int w = e ? atoi(e) : 8;
    return w >= 16 ? 16 : (w >= 8 ? 8 : 4);
  }();
  static const int v64 = [] {
    const char *e = tune_env("SF_THIN_WAVES64");
    const int w = e ? atoi(e) : 8;
    return w >= 16 ? 16 : (w >= 8 ? 8 : 4);
  }();
  return C == 32 ? v32 : (C == 64 ? v64 : 16);
}

constexpr int kThinMaxTiles = 64;   // 32-position tiles per workgroup (RW <= 2048; 512 above 8 channels)
constexpr int kThinMaxG = 16;

// eight consecutive channels of one position
template <typename T> struct K8;
template <> struct K8<bf16> {
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
  __device__ __forceinline__ static K8 load(const bf16 *p) {
    K8 r;
    r.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(p));
    return r;
  }
  __device__ __forceinline__ void store(bf16 *p) const { *reinterpret_cast<u32x4 *>(p) = __builtin_bit_cast(u32x4, v); }
  __device__ __forceinline__ static K8 zero() {
    K8 r;
    u32x4 z = {0u, 0u, 0u, 0u};
    r.v = __builtin_bit_cast(bf16x8, z);
    return r;
  }
};
template <> struct K8<f16> {
  f16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
  __device__ __forceinline__ static K8 load(const f16 *p) {
    K8 r;
    r.v = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(p));
    return r;
  }
  __device__ __forceinline__ void store(f16 *p) const { *reinterpret_cast<u32x4 *>(p) = __builtin_bit_cast(u32x4, v); }
  __device__ __forceinline__ static K8 zero() {
    K8 r;
    u32x4 z = {0u, 0u, 0u, 0u};
    r.v = __builtin_bit_cast(f16x8, z);
    return r;
  }
};
template <> struct K8<float> {
  f32x4 v[2];
  __device__ __forceinline__ float get(int i) const { return v[i >> 2][i & 3]; }
  __device__ __forceinline__ void set(int i, float x) { v[i >> 2][i & 3] = x; }
  __device__ __forceinline__ static K8 load(const float *p) {
    K8 r;
    r.v[0] = *reinterpret_cast<const f32x4 *>(p);
    r.v[1] = *reinterpret_cast<const f32x4 *>(p + 4);
    return r;
  }
  __device__ __forceinline__ void store(float *p) const {
    *reinterpret_cast<f32x4 *>(p) = v[0];
    *reinterpret_cast<f32x4 *>(p + 4) = v[1];
  }
  __device__ __forceinline__ static K8 zero() {
    K8 r;
    r.v[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.v[1] = r.v[0];
    return r;
  }
};

__device__ __forceinline__ void mma_step(f32x16 &acc, const K8<bf16> &a, const K8<bf16> &b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(f32x16 &acc, const K8<f16> &a, const K8<f16> &b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(f32x16 &acc, const K8<float> &a, const K8<float> &b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.get(j), b.get(j), acc, 0, 0, 0);
}

// fp32x (the split-operand mode of the fp32 engine, common.h X3P<X3_F16>): eight consecutive channels as (hi, lo') fp16 halves.  The staged
// rows and the staged weights are split ONCE on their way into LDS (two fp16 images in the bytes of the fp32 one); operands that come
// from registers or straight from global memory (weights of conv_thin, second source, an accumulator fed back) are split where used.
struct K8x {
  f16x8 h, l;
  __device__ __forceinline__ static K8x from(const K8<float> &v) {
    K8x r;
    x3_split<X3_F16>(v.v[0], v.v[1], r.h, r.l);
    return r;
  }
  __device__ __forceinline__ static K8x load(const f16 *ph, const f16 *pl) {
    K8x r;
    r.h = *reinterpret_cast<const f16x8 *>(ph);
    r.l = *reinterpret_cast<const f16x8 *>(pl);
    return r;
  }
  __device__ __forceinline__ void store(f16 *ph, f16 *pl) const {
    *reinterpret_cast<f16x8 *>(ph) = h;
    *reinterpret_cast<f16x8 *>(pl) = l;
  }
  __device__ __forceinline__ static K8x zero() {
    K8x r;
    const u32x4 z = {0u, 0u, 0u, 0u};
    r.h = __builtin_bit_cast(f16x8, z);
    r.l = r.h;
    return r;
  }
};
__device__ __forceinline__ void mma_step_x3(f32x16 &accM, f32x16 &accL, const K8x &a, const K8x &b) { x3_mfma<X3_F16>(a.h, a.l, b.h, b.l, accM, accL); }
__device__ __forceinline__ void x3_fold(f32x16 &accM, const f32x16 &accL) {
#pragma unroll
  for (int i = 0; i < 16; ++i) accM[i] = fmaf(accL[i], X3P<X3_F16>::INV, accM[i]);
}

// Sum over the 32 lanes of each half-wave with DPP row operations (VALU only: no LDS crossbar traffic, unlike
// __shfl_xor = ds_bpermute); every lane receives the total of its own half.  Fixed order -> deterministic.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float2 half_sums(float v) {   // (total of lanes 0-31, total of lanes 32-63), on every lane
  v += dpp_mov<0xb1, 0xf>(v);
  v += dpp_mov<0x4e, 0xf>(v);
  v += dpp_mov<0x124, 0xf>(v);
  v += dpp_mov<0x128, 0xf>(v);
  v += dpp_mov<0x142, 0xa>(v);
  return make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)),
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)));
}
__device__ __forceinline__ float half_sum32(float v, int half) {
  v += dpp_mov<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x124, 0xf>(v);   // row_ror:4
  v += dpp_mov<0x128, 0xf>(v);   // row_ror:8      -> every lane holds the sum of its row of 16
  v += dpp_mov<0x142, 0xa>(v);   // row_bcast:15   -> rows 1 and 3 add the total of the row before
  const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return half ? hi : lo;
}

// (mean, rstd) of one (clip, group) from its nch chunk partials, any nch: lane j of the half-wave folds partials
// j, j+32, ... in order, then the fixed shuffle tree of gn_merge32 -- deterministic.
template <bool FAST = false>
__device__ __forceinline__ float2 gn_merge_n(const float *__restrict__ sl, int G, int nch, int chunk_rows, int L, int cpg, float eps,
                                             int lane32) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int i = lane32; i < nch; i += 32) {
    const int rows = min(chunk_rows, L - i * chunk_rows);
    welford_merge_t<FAST>(n, mean, m2, (float)rows * (float)cpg, sl[(size_t)i * G * 2], sl[(size_t)i * G * 2 + 1]);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const float nb = __shfl_down(n, off, 32), mb = __shfl_down(mean, off, 32), qb = __shfl_down(m2, off, 32);
    welford_merge_t<FAST>(n, mean, m2, nb, mb, qb);
  }
  const float mu = __shfl(mean, 0, 32), var = div_t<FAST>(__shfl(m2, 0, 32), __shfl(n, 0, 32));
  return make_float2(mu, rsqrtf(var + eps));
}

// four consecutive channels of one position (epilogue granularity)
template <typename T> struct K4;
template <> struct K4<bf16> {
  uint2 raw;
  __device__ __forceinline__ float get(int i) const {
    const unsigned int w = i < 2 ? raw.x : raw.y;
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  __device__ __forceinline__ static K4 load(const bf16 *p) {
    K4 r;
    r.raw = *reinterpret_cast<const uint2 *>(p);
    return r;
  }
  __device__ __forceinline__ static K4 zero() {
    K4 r;
    r.raw = make_uint2(0u, 0u);
    return r;
  }
};
template <> struct K4<f16> {
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  f16x4 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ static K4 load(const f16 *p) {
    K4 r;
    r.v = *reinterpret_cast<const f16x4 *>(p);
    return r;
  }
  __device__ __forceinline__ static K4 zero() {
    K4 r;
    r.v = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    return r;
  }
};
template <> struct K4<float> {
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ static K4 load(const float *p) {
    K4 r;
    r.v = *reinterpret_cast<const f32x4 *>(p);
    return r;
  }
  __device__ __forceinline__ static K4 zero() {
    K4 r;
    r.v = f32x4{0.f, 0.f, 0.f, 0.f};
    return r;
  }
};

// CIN: channels of a staged source row; NOUT: output channels (8, 32, 64); C2: channels of the concatenated second source
// (the split mode carries a second accumulator set: its instantiations are built for at most 8 waves, 256 registers each)
template <typename T, int CIN, int TAPS, int C2, int PRO, int NOUT, bool X3 = false>
__global__ __launch_bounds__(X3 ? 512 : 1024) void conv_thin_kernel(const ConvThinArgs a) {
  static_assert(!X3 || sizeof(T) == 4, "split mode: fp32 activations");
  constexpr int E = 8;
  constexpr int QC = CIN / E;               // channel octets of a staged row
  constexpr int S1 = TAPS * QC, S2 = C2 / E, S = S1 + S2, NSTEP = (S + 1) / 2;
  constexpr int NL = S1 / 2;                // steps whose two slots both come from the staged source
  constexpr bool MIX = (S1 & 1) != 0;       // step NL: lanes 0-31 staged source, lanes 32-63 second source / nothing
  constexpr int NS2 = NSTEP - NL;           // steps that touch the second source (incl. the mixed one)
  constexpr int NCB = (NOUT + 31) / 32;     // 32-wide blocks of output channels
  constexpr int SROW = CIN + E;             // LDS row pitch in elements (16-byte skew against bank conflicts)
  constexpr int HALO = TAPS / 2;
  constexpr int NV = 6;                     // staged 8-channel vectors per thread (upper bound, see thin_go)
  constexpr bool FAST = !std::is_same<T, float>::value;
  constexpr bool KEEP_W = sizeof(T) == 2;   // bf16: the wave's weight fragments stay in registers
  static_assert(PRO == 0 || CIN == NOUT, "prologues belong to the item convolutions (C -> C)");

  extern __shared__ __align__(16) unsigned char smem[];
  float *sc = reinterpret_cast<float *>(smem);
  float *sh = sc + CIN;
  float *ep = sh + CIN;     // epilogue vectors: bias | per-clip scale | per-clip add, 64 floats each (fetched up front)
  float *part = ep + 192;   // [tile][G][2]
  T *tile = reinterpret_cast<T *>(part + kThinMaxTiles * kThinMaxG * 2);
  // split mode: the staged rows as two fp16 images in the same bytes (thin_lds_bytes reserves (rw >> up_shift) + 2 HALO + 2 rows)
  f16 *tileH = reinterpret_cast<f16 *>(tile);
  f16 *tileL = tileH + (size_t)((a.rw >> a.up_shift) + 2 * HALO + 2) * SROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.x / a.nchw, ch = blockIdx.x - b * a.nchw;
  const int r0 = ch * a.rw;                       // first output position of the workgroup
  const int rows = min(a.rw, a.L - r0);
  const int ntile = (rows + 31) >> 5;
  const int K = S * E;
  const int cpg = NOUT / a.G;
  const T *src = static_cast<const T *>(a.src);
  const T *wgt = static_cast<const T *>(a.w);
  const int cb = wave % NCB;   // launch guarantees NW % NCB == 0: a wave keeps one block of output channels
  // source rows feeding positions [r0 - HALO, r0 + rows + HALO): nearest-neighbour upsampling reads row p >> up_shift
  const int j0 = (r0 - HALO) >> a.up_shift;       // arithmetic shift: -1 stays -1 (the zero padding row)
  const int nsrc = ((r0 + rows - 1 + HALO) >> a.up_shift) - j0 + 1;

  // ---- every global read of the workgroup is issued up front: ONE memory round trip ---------------------------
  // (a) the staged source rows
  const int total = nsrc * QC;
  K8<T> xin[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * blockDim.x;
    const int rr = idx / QC, q = idx - rr * QC;
    const int j = j0 + rr;
    const bool ok = idx < total && j >= 0 && j < a.Ls;
    xin[i] = ok ? K8<T>::load(src + ((size_t)b * a.Ls + j) * a.src_ld + q * E) : K8<T>::zero();
  }
  // (b) the wave's weights: lane = output channel cb*32 + l32, slot 2s + half
  K8<T> wf[KEEP_W ? NSTEP : 1];
  const int wc = cb * 32 + l32;
  const T *wrow = wgt + (size_t)(wc < NOUT ? wc : 0) * K + half * E;
  auto wload = [&](int s) { return (wc < NOUT && 2 * s + half < S) ? K8<T>::load(wrow + s * 2 * E) : K8<T>::zero(); };
  if constexpr (KEEP_W) {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wf[s] = wload(s);
  }
  // (c) second source and residual of the wave's first tile
  const int t_first = wave / NCB;
  const int rowl_first = t_first * 32 + l32;
  const bool rv_first = t_first < ntile && rowl_first < rows;
  const size_t grow_first = (size_t)b * a.L + r0 + (rv_first ? rowl_first : 0);
  K8<T> s2f[NS2 > 0 ? NS2 : 1];
  auto s2load = [&](int i, bool rv, size_t grow) {
    const int q2 = 2 * (NL + i) + half - S1;
    return (rv && q2 >= 0 && q2 < S2) ? K8<T>::load(static_cast<const T *>(a.src2) + grow * a.src2_ld + q2 * E) : K8<T>::zero();
  };
  if constexpr (S2 > 0) {
#pragma unroll
    for (int i = 0; i < NS2; ++i) s2f[i] = s2load(i, rv_first, grow_first);
  } else if constexpr (NS2 > 0) {
    s2f[0] = K8<T>::zero();
  }
  K4<T> resf[4];
  const bool res_g = a.res != nullptr && !a.res_self;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const bool vv = NOUT >= 32 || 8 * v < NOUT;
    resf[v] = (vv && res_g && rv_first) ? K4<T>::load(static_cast<const T *>(a.res) + grow_first * a.res_ld + cb * 32 + half * 4 + 8 * v)
                                        : K4<T>::zero();
  }

  // ---- epilogue vectors and prologue parameters -> LDS (their loads ride in the same round trip as the rows above) ----
  if (tid < 64) {
    const int c = min(tid, NOUT - 1);
    ep[tid] = a.bias ? a.bias[c] : 0.f;
    ep[64 + tid] = a.bscale ? a.bscale[(size_t)b * a.bscale_ld + c] : 1.f;
    ep[128 + tid] = a.badd ? a.badd[(size_t)b * a.badd_ld + c] : 0.f;
  }
  if constexpr (PRO == 0) __syncthreads();
  if constexpr (PRO == 1) {
    for (int g = tid >> 5; g < a.G; g += blockDim.x >> 5) {
      const float2 st = gn_merge_n<FAST>(a.stats_in + ((size_t)b * a.nch_in * a.G + g) * 2, a.G, a.nch_in, a.chunk_in, a.L, cpg, a.eps, l32);
      if (l32 < cpg) {
        const int c = g * cpg + l32;
        const float s = st.y * a.gamma[c];
        sc[c] = s;
        sh[c] = a.beta[c] - st.x * s;
      }
    }
    __syncthreads();
  } else if constexpr (PRO == 2) {
    if (tid < CIN) {
      sc[tid] = a.ss ? 1.0f + a.ss[(size_t)b * a.ss_ld + tid] : 1.0f;
      sh[tid] = a.ss ? a.ss[(size_t)b * a.ss_ld + CIN + tid] : 0.0f;
    }
    __syncthreads();
  }

  // ---- prologue applied once per element, staged in LDS ---------------------------------------------------------
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * blockDim.x;
    if (idx < total) {
      const int rr = idx / QC, q = idx - rr * QC;
      K8<T> v = xin[i];
      if constexpr (PRO == 1) {
        const int j = j0 + rr;
        const bool ok = j >= 0 && j < a.Ls;   // the convolution pads the ACTIVATED tensor with zeros
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float y = fmaf(v.get(e), sc[q * E + e], sh[q * E + e]);
          v.set(e, ok ? silu_t<FAST>(y) : 0.f);
        }
      } else if constexpr (PRO == 2) {
        // LayerNorm over the channels of the row: its QC octets sit on QC consecutive lanes
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) sum += v.get(e);
#pragma unroll
        for (int o = QC >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)CIN;
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float d = v.get(e) - mean;
          sq = fmaf(d, d, sq);
        }
#pragma unroll
        for (int o = QC >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq / (float)CIN + a.eps);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float y = (v.get(e) - mean) * rstd;
          v.set(e, fmaf(y, sc[q * E + e], sh[q * E + e]));
        }
      }
      if constexpr (X3) K8x::from(v).store(tileH + rr * SROW + q * E, tileL + rr * SROW + q * E);
      else v.store(tile + rr * SROW + q * E);
    }
  }
  __syncthreads();

  // ---- tiles ---------------------------------------------------------------------------------------------------
  for (int item = wave; item < ntile * NCB; item += NW) {
    const int t = item / NCB;
    const int row_l = t * 32 + l32;
    const bool rvalid = row_l < rows;
    const size_t grow = (size_t)b * a.L + r0 + (rvalid ? row_l : 0);
    if (item != wave) {   // later tiles of this wave: fetch their second source / residual now
      if constexpr (S2 > 0) {
#pragma unroll
        for (int i = 0; i < NS2; ++i) s2f[i] = s2load(i, rvalid, grow);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bool vv = NOUT >= 32 || 8 * v < NOUT;
        resf[v] = (vv && res_g && rvalid) ? K4<T>::load(static_cast<const T *>(a.res) + grow * a.res_ld + cb * 32 + half * 4 + 8 * v)
                                          : K4<T>::zero();
      }
    }
    const int prow = r0 + (rvalid ? row_l : 0) - HALO;   // upsampled position of tap 0
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (X3) {
      f32x16 accL;
#pragma unroll
      for (int i = 0; i < 16; ++i) accL[i] = 0.f;
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        K8x bf;
        if (s < NL || (MIX && s == NL)) {
          const int slot = min(2 * s + half, S1 - 1);
          const int tap = slot / QC, q = slot - tap * QC;
          const int jj = ((prow + tap) >> a.up_shift) - j0;
          bf = K8x::load(tileH + jj * SROW + q * E, tileL + jj * SROW + q * E);
          if (MIX && s == NL) {
            const K8x s2 = K8x::from(s2f[0]);
            if (half) bf = s2;
          }
        } else {
          bf = K8x::from(s2f[NS2 > 0 ? s - NL : 0]);
        }
        mma_step_x3(acc, accL, K8x::from(wload(s)), bf);
      }
      x3_fold(acc, accL);
    } else {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      K8<T> bf;
      if (s < NL || (MIX && s == NL)) {   // staged source: slot -> (tap, octet); row = (position + tap) >> up_shift
        const int slot = min(2 * s + half, S1 - 1);
        const int tap = slot / QC, q = slot - tap * QC;
        const int jj = ((prow + tap) >> a.up_shift) - j0;
        bf = K8<T>::load(tile + jj * SROW + q * E);
        if (MIX && s == NL) {
          if (half) bf = s2f[0];
        }
      } else {
        bf = s2f[NS2 > 0 ? s - NL : 0];
      }
      if constexpr (KEEP_W) mma_step(acc, wf[s], bf);
      else mma_step(acc, wload(s), bf);
    }
    }

    // ---- epilogue: lane = position row_l, registers 4v..4v+3 = channels cb*32 + half*4 + 8v + {0..3} -----------
    float xs[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (!(NOUT >= 32 || 8 * v < NOUT)) continue;
      const int c0 = cb * 32 + half * 4 + 8 * v;
      float val[4];
      const f32x4 bias = *reinterpret_cast<const f32x4 *>(ep + c0), bs = *reinterpret_cast<const f32x4 *>(ep + 64 + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = (acc[4 * v + e] + bias[e]) * bs[e];
      if (a.res_self) {   // residual = the staged (modulated) input itself (CIN == NOUT, no upsampling)
        if constexpr (X3) {   // (reassembled from its two halves: 22 significant bits)
          typedef f16 f16x4_t __attribute__((ext_vector_type(4)));
          const f16x4_t rh = *reinterpret_cast<const f16x4_t *>(tileH + (row_l + HALO) * SROW + c0);
          const f16x4_t rl = *reinterpret_cast<const f16x4_t *>(tileL + (row_l + HALO) * SROW + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] += fmaf((float)rl[e], X3P<X3_F16>::INV, (float)rh[e]);
        } else {
          const K4<T> rp = K4<T>::load(tile + (row_l + HALO) * SROW + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] += rp.get(e);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] += resf[v].get(e);
      }
      {
        const f32x4 ba = *reinterpret_cast<const f32x4 *>(ep + 128 + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] += ba[e];
      }
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = from_f<T>(val[e]);
      if (rvalid) {
        T *op = static_cast<T *>(a.out) + grow * a.out_ld + c0;
        if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2 *>(op) = *reinterpret_cast<const uint2 *>(o);
        else *reinterpret_cast<f32x4 *>(op) = *reinterpret_cast<const f32x4 *>(o);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) xs[v][e] = rvalid ? to_f(o[e]) : 0.f;   // statistics see the STORED values
    }
    // GroupNorm partial of this tile: two passes over registers (sum -> mean, then centred squares); the sums over
    // the 32 positions are DPP reductions (plus the other half-wave when a group spans both: cpg == 8).
    if (a.stats_out) {
      const int vrows = min(32, rows - t * 32);
      if constexpr (NOUT == 8) {   // one channel per group (G == 8): this lane's four channels are four groups
        const float cnt = (float)vrows;
        float m[4], q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = div_t<FAST>(half_sum32(xs[0][e], half), cnt);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = rvalid ? xs[0][e] - m[e] : 0.f;
          q[e] = half_sum32(d * d, half);
        }
        if (l32 == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float *pp = part + ((size_t)t * a.G + half * 4 + e) * 2;
            pp[0] = m[e];
            pp[1] = q[e];
          }
        }
      } else {
        const float cnt = (float)vrows * (float)cpg;
        float gs[4], gq[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float2 hs = half_sums((xs[v][0] + xs[v][1]) + (xs[v][2] + xs[v][3]));
          const float sum = cpg == 8 ? hs.x + hs.y : (half ? hs.y : hs.x);
          gs[v] = div_t<FAST>(sum, cnt);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float q = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = rvalid ? xs[v][e] - gs[v] : 0.f;
            q = fmaf(d, d, q);
          }
          const float2 hq = half_sums(q);
          gq[v] = cpg == 8 ? hq.x + hq.y : (half ? hq.y : hq.x);
        }
        if (l32 == 0 && (cpg == 4 || half == 0)) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int g = (cb * 32 + half * 4 + 8 * v) / cpg;
            float *pp = part + ((size_t)t * a.G + g) * 2;
            pp[0] = gs[v];
            pp[1] = gq[v];
          }
        }
      }
    }
  }

  // ---- (mean, M2) of this workgroup's output per group, tiles folded in order ------------------------------------
  if (a.stats_out) {
    __syncthreads();
    // one wave per group: lane t holds tile t's (count, mean, M2); exact pooled statistics in two DPP-reduced passes
    for (int g = wave; g < a.G; g += NW) {
      const bool on = lane < ntile;
      const float *pp = part + ((size_t)(on ? lane : 0) * a.G + g) * 2;
      const float n_t = on ? (float)min(32, rows - lane * 32) * (float)cpg : 0.f;
      const float m_t = on ? pp[0] : 0.f, q_t = on ? pp[1] : 0.f;
      const float2 sn = half_sums(n_t), sm = half_sums(n_t * m_t);
      const float n = sn.x + sn.y, mean = div_t<FAST>(sm.x + sm.y, n);
      const float d = m_t - mean;
      const float2 sq = half_sums(fmaf(n_t * d, d, q_t));
      if (lane == 0) {
        float *so = a.stats_out + (((size_t)b * a.nchw + ch) * a.G + g) * 2;
        so[0] = mean;
        so[1] = sq.x + sq.y;
      }
    }
  }
}

template <typename T> size_t thin_lds_bytes(int cin, int taps, int rw, int up_shift) {
  const int halo = taps / 2;
  const size_t rows = (size_t)(rw >> up_shift) + 2 * halo + 2;   // source rows a workgroup stages (+ slack)
  return (size_t)(2 * cin + 192 + kThinMaxTiles * kThinMaxG * 2) * sizeof(float) + rows * (cin + 8) * sizeof(T);
}

template <typename T, int CIN, int TAPS, int C2, int PRO, int NOUT, bool X3 = false> hipError_t thin_go(const ConvThinArgs &a, hipStream_t s) {
  const size_t lds = thin_lds_bytes<T>(CIN, TAPS, a.rw, a.up_shift);
  auto kern = conv_thin_kernel<T, CIN, TAPS, C2, PRO, NOUT, X3>;
  if (lds > 64 * 1024) {
    static bool raised = false;   // per instantiation
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      raised = true;
    }
  }
  constexpr int NCB = (NOUT + 31) / 32;
  int nw = ((a.rw + 31) / 32) * NCB;
  if (nw > thin_max_waves(NOUT)) nw = thin_max_waves(NOUT);
  if (X3 && nw > 8) nw = 8;
  nw = (nw / NCB) * NCB;
  // staging registers: the source rows of a workgroup (<= rw + 2) * CIN/8 vectors must fit NV = 6 per thread
  if (((a.rw >> a.up_shift) + 3) * (CIN / 8) > 6 * nw * 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(a.B * a.nchw), dim3(nw * 64), lds, s, a);
  return hipGetLastError();
}

// the instantiated shapes: (CIN, taps, C2, pro, NOUT)
struct ThinShape {
  int cin, taps, c2, pro, nout;
};
constexpr ThinShape kThinShapes[] = {
    // ResnetItem convolutions (GroupNorm+SiLU in) and Modulation+InjectChannels (LayerNorm-modulate in)
    {8, 3, 0, 1, 8},   {32, 3, 0, 1, 32},  {64, 3, 0, 1, 64},
    {8, 1, 8, 2, 8},   {32, 1, 32, 2, 32}, {64, 1, 32, 2, 64}, {64, 1, 64, 2, 64},
    // patchify down convolutions on the (rows/f, f*cin) view
    {32, 1, 0, 0, 32}, {64, 1, 0, 0, 64},  {128, 1, 0, 0, 64},
    // nearest-upsample + conv3 up convolutions
    {32, 3, 0, 0, 8},  {64, 3, 0, 0, 32},  {64, 3, 0, 0, 64},
};

// the split-operand instantiations (fp32x engine): the MFMA levels (32 / 64 channels); the 8-channel level stays on the vector kernels
static hipError_t thin_dispatch_x3(const ConvThinArgs &a, hipStream_t s) {
#define SF_THINX(CC, TT, C22, PP, NN) \
  if (a.C == CC && a.taps == TT && a.C2 == C22 && a.pro == PP && a.N == NN) return thin_go<float, CC, TT, C22, PP, NN, true>(a, s)
  SF_THINX(32, 3, 0, 1, 32);
  SF_THINX(64, 3, 0, 1, 64);
  SF_THINX(32, 1, 32, 2, 32);
  SF_THINX(64, 1, 32, 2, 64);
  SF_THINX(64, 1, 64, 2, 64);
  SF_THINX(32, 1, 0, 0, 32);
  SF_THINX(64, 1, 0, 0, 64);
  SF_THINX(128, 1, 0, 0, 64);
  SF_THINX(64, 3, 0, 0, 32);
  SF_THINX(64, 3, 0, 0, 64);
#undef SF_THINX
  return hipErrorInvalidValue;
}

template <typename T> hipError_t thin_dispatch(const ConvThinArgs &a, hipStream_t s) {
#define SF_THIN(CC, TT, C22, PP, NN) \
  if (a.C == CC && a.taps == TT && a.C2 == C22 && a.pro == PP && a.N == NN) return thin_go<T, CC, TT, C22, PP, NN>(a, s)
  SF_THIN(8, 3, 0, 1, 8);
  SF_THIN(32, 3, 0, 1, 32);
  SF_THIN(64, 3, 0, 1, 64);
  SF_THIN(8, 1, 8, 2, 8);
  SF_THIN(32, 1, 32, 2, 32);
  SF_THIN(64, 1, 32, 2, 64);
  SF_THIN(64, 1, 64, 2, 64);
  SF_THIN(32, 1, 0, 0, 32);
  SF_THIN(64, 1, 0, 0, 64);
  SF_THIN(128, 1, 0, 0, 64);
  SF_THIN(32, 3, 0, 0, 8);
  SF_THIN(64, 3, 0, 0, 32);
  SF_THIN(64, 3, 0, 0, 64);
#undef SF_THIN
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------
// Item tail in ONE launch:  y = x + Conv3(SiLU(GN2(h)));  m = LN_C(y) * (1 + scale) + shift;  z = m + Conv1x1(cat[m, ctx]) (+ badd)
// (the second half of a ResnetItem, the ModulationItem and the InjectChannelsItem).  y and m never leave the registers:
// the conv2 accumulator of a wave holds ALL C channels of its 32 positions (channel 8v + 4*half + e of block cb in
// register 4v + e), LayerNorm is a per-lane sum plus one exchange with the other half-wave, and the modulated tile is fed
// straight back as the B operand of the 1x1 convolution -- an accumulator used as an operand carries its channels in
// the order 16s + 8*(j >> 2) + 4*half + (j & 3), so the weight rows (A operand, staged in LDS) are read with the same
// permutation.  The context channels follow as ordinary 8-channel slots.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int C, int C2, bool X3 = false>
__global__ __launch_bounds__(X3 ? 512 : 1024) void thin_tail_kernel(const ThinTailArgs a) {
  static_assert(!X3 || sizeof(T) == 4, "split mode: fp32 activations");
  constexpr int E = 8;
  constexpr int QC = C / E;
  constexpr int S2 = 3 * QC, NST2 = (S2 + 1) / 2;     // conv2: slots / MFMA steps
  constexpr int NSTM = (C + 15) / 16;                 // inject, modulated part: 16 channels per step
  constexpr int SC = C2 / E, NSTC = (SC + 1) / 2;     // inject, context part
  constexpr int NCB = (C + 31) / 32;
  constexpr int K2 = 3 * C, K3 = C + C2;
  constexpr int P2 = K2 + E, P3 = K3 + E;             // LDS row pitches of the staged weights (16-byte skew)
  constexpr int SROW = C + E;
  constexpr int NV = 6;
  constexpr bool FAST = !std::is_same<T, float>::value;

  extern __shared__ __align__(16) unsigned char smem[];
  float *sc = reinterpret_cast<float *>(smem);        // GroupNorm scale / shift per channel
  float *sh = sc + C;
  float *lsc = sh + C;                                // 1 + modulation scale, modulation shift
  float *lsh = lsc + C;
  float *ep = lsh + C;                                // bias2 | bias3 + per-clip add, 64 floats each
  float *part = ep + 128;                             // [tile][G][2]
  T *w2s = reinterpret_cast<T *>(part + kThinMaxTiles * kThinMaxG * 2);
  T *w3s = w2s + C * P2;
  T *tile = w3s + C * P3;
  // split mode: each staged fp32 region holds two fp16 images (hi, lo') instead
  f16 *w2H = reinterpret_cast<f16 *>(w2s), *w2L = w2H + C * P2;
  f16 *w3H = reinterpret_cast<f16 *>(w3s), *w3L = w3H + C * P3;
  f16 *tileH = reinterpret_cast<f16 *>(tile), *tileL = tileH + (size_t)(a.rw + 4) * SROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.x / a.nchw, ch = blockIdx.x - b * a.nchw;
  const int r0 = ch * a.rw;
  const int rows = min(a.rw, a.L - r0);
  const int ntile = (rows + 31) >> 5;
  const int cpg = C / a.G;
  const T *src = static_cast<const T *>(a.h);

  // ---- all global reads up front --------------------------------------------------------------------------------
  const int total = (rows + 2) * QC;
  K8<T> xin[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * blockDim.x;
    const int rr = idx / QC, q = idx - rr * QC;
    const int pos = r0 - 1 + rr;
    const bool ok = idx < total && pos >= 0 && pos < a.L;
    xin[i] = ok ? K8<T>::load(src + ((size_t)b * a.L + pos) * C + q * E) : K8<T>::zero();
  }
  // weights -> LDS (rows of the packed [C][K] matrices, 8 elements per thread and pass)
  for (int i = tid; i < C * (K2 / E); i += blockDim.x) {
    const int r = i / (K2 / E), v = i - r * (K2 / E);
    const K8<T> w = K8<T>::load(static_cast<const T *>(a.w2) + (size_t)r * K2 + v * E);
    if constexpr (X3) K8x::from(w).store(w2H + r * P2 + v * E, w2L + r * P2 + v * E);
    else w.store(w2s + r * P2 + v * E);
  }
  for (int i = tid; i < C * (K3 / E); i += blockDim.x) {
    const int r = i / (K3 / E), v = i - r * (K3 / E);
    const K8<T> w = K8<T>::load(static_cast<const T *>(a.w3) + (size_t)r * K3 + v * E);
    if constexpr (X3) K8x::from(w).store(w3H + r * P3 + v * E, w3L + r * P3 + v * E);
    else w.store(w3s + r * P3 + v * E);
  }
  // GroupNorm statistics of h, modulation vectors
  for (int g = tid >> 5; g < a.G; g += blockDim.x >> 5) {
    const float2 st = gn_merge_n<FAST>(a.stats_in + ((size_t)b * a.nch_in * a.G + g) * 2, a.G, a.nch_in, a.chunk_in, a.L, cpg, a.eps_gn, l32);
    if (l32 < cpg) {
      const int c = g * cpg + l32;
      const float s = st.y * a.gamma[c];
      sc[c] = s;
      sh[c] = a.beta[c] - st.x * s;
    }
  }
  if (tid < C) {
    lsc[tid] = 1.0f + a.ss[(size_t)b * a.ss_ld + tid];
    lsh[tid] = a.ss[(size_t)b * a.ss_ld + C + tid];
    ep[tid] = a.bias2[tid];
    ep[64 + tid] = a.bias3[tid] + (a.badd ? a.badd[(size_t)b * a.badd_ld + tid] : 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * blockDim.x;
    if (idx < total) {
      const int rr = idx / QC, q = idx - rr * QC;
      const int pos = r0 - 1 + rr;
      const bool ok = pos >= 0 && pos < a.L;
      K8<T> v = xin[i];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float y = fmaf(v.get(e), sc[q * E + e], sh[q * E + e]);
        v.set(e, ok ? silu_t<FAST>(y) : 0.f);
      }
      if constexpr (X3) K8x::from(v).store(tileH + rr * SROW + q * E, tileL + rr * SROW + q * E);
      else v.store(tile + rr * SROW + q * E);
    }
  }
  __syncthreads();

  for (int t = wave; t < ntile; t += NW) {
    const int row_l = t * 32 + l32;
    const bool rvalid = row_l < rows;
    const size_t grow = (size_t)b * a.L + r0 + (rvalid ? row_l : 0);
    // context fragments of this tile (second source of the 1x1 convolution): issue now, use last
    K8<T> cf[NSTC > 0 ? NSTC : 1];
#pragma unroll
    for (int i = 0; i < NSTC; ++i) {
      const int q2 = 2 * i + half;
      cf[i] = (rvalid && q2 < SC) ? K8<T>::load(static_cast<const T *>(a.ctx) + grow * a.ctx_ld + q2 * E) : K8<T>::zero();
    }
    K4<T> rxf[NCB][4];   // residual x of this tile, issued with the context fragments
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c0 = cb * 32 + half * 4 + 8 * v;
        rxf[cb][v] = (c0 < C && rvalid) ? K4<T>::load(static_cast<const T *>(a.x) + grow * C + c0) : K4<T>::zero();
      }
    // ---- conv2 over the staged SiLU(GN2(h)) ----------------------------------------------------------------------
    f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    const int prow = rvalid ? row_l : 0;
    if constexpr (X3) {
      f32x16 accL[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) accL[cb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NST2; ++s) {
        const int slot = min(2 * s + half, S2 - 1);
        const bool sv = 2 * s + half < S2;
        const int tap = slot / QC, q = slot - tap * QC;
        K8x bf = K8x::load(tileH + (prow + tap) * SROW + q * E, tileL + (prow + tap) * SROW + q * E);
        if (!sv) bf = K8x::zero();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int c = cb * 32 + l32;
          K8x af = K8x::load(w2H + min(c, C - 1) * P2 + slot * E, w2L + min(c, C - 1) * P2 + slot * E);
          if (c >= C || !sv) af = K8x::zero();
          mma_step_x3(acc[cb], accL[cb], af, bf);
        }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) x3_fold(acc[cb], accL[cb]);
    } else {
#pragma unroll
    for (int s = 0; s < NST2; ++s) {
      const int slot = min(2 * s + half, S2 - 1);
      const bool sv = 2 * s + half < S2;
      const int tap = slot / QC, q = slot - tap * QC;
      K8<T> bf = K8<T>::load(tile + (prow + tap) * SROW + q * E);
      if (!sv) bf = K8<T>::zero();
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int c = cb * 32 + l32;
        K8<T> af = K8<T>::load(w2s + min(c, C - 1) * P2 + slot * E);
        if (c >= C || !sv) af = K8<T>::zero();
        mma_step(acc[cb], af, bf);
      }
    }
    }
    // ---- y = conv2 + bias + x;  LayerNorm over the C channels of the position; modulate ------------------------------
    float sum = 0.f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c0 = cb * 32 + half * 4 + 8 * v;
        if (c0 < C) {
          const f32x4 bias = *reinterpret_cast<const f32x4 *>(ep + c0);
          const K4<T> rx = rxf[cb][v];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = to_f(from_f<T>(acc[cb][4 * v + e] + bias[e] + rx.get(e)));   // as stored by the unfused path
            acc[cb][4 * v + e] = y;
            sum += y;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[cb][4 * v + e] = 0.f;
        }
      }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (cb * 32 + half * 4 + 8 * v < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = acc[cb][4 * v + e] - mean;
            sq = fmaf(d, d, sq);
          }
        }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq / (float)C + a.eps_ln);
    // m, rounded to the compute type (it is both the operand and the residual of the 1x1 convolution)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c0 = cb * 32 + half * 4 + 8 * v;
        if (c0 < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[cb][4 * v + e] = to_f(from_f<T>(fmaf((acc[cb][4 * v + e] - mean) * rstd, lsc[c0 + e], lsh[c0 + e])));
        }
      }
    // ---- z = m + W3 . [m | ctx] -------------------------------------------------------------------------------------
    f32x16 zc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) zc[cb][i] = 0.f;
    if constexpr (X3) {
      typedef f16 f16x4_t __attribute__((ext_vector_type(4)));
      f32x16 zl[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int i = 0; i < 16; ++i) zl[cb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NSTM; ++s) {
        K8<float> bf32;
#pragma unroll
        for (int j = 0; j < 8; ++j) bf32.set(j, acc[s >> 1][8 * (s & 1) + j]);
        const K8x bf = K8x::from(bf32);   // the modulated tile, split in registers
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int c = cb * 32 + l32;
          const int o = min(c, C - 1) * P3 + 16 * s + 4 * half;
          const f16x4_t h0 = *reinterpret_cast<const f16x4_t *>(w3H + o), h1 = *reinterpret_cast<const f16x4_t *>(w3H + o + 8);
          const f16x4_t l0 = *reinterpret_cast<const f16x4_t *>(w3L + o), l1 = *reinterpret_cast<const f16x4_t *>(w3L + o + 8);
          K8x af;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool k0 = c < C && 16 * s + 4 * half + j < C, k1 = c < C && 16 * s + 8 + 4 * half + j < C;
            af.h[j] = k0 ? h0[j] : (f16)0.f;
            af.l[j] = k0 ? l0[j] : (f16)0.f;
            af.h[4 + j] = k1 ? h1[j] : (f16)0.f;
            af.l[4 + j] = k1 ? l1[j] : (f16)0.f;
          }
          mma_step_x3(zc[cb], zl[cb], af, bf);
        }
      }
#pragma unroll
      for (int i = 0; i < NSTC; ++i) {
        const int q2 = 2 * i + half;
        const K8x cx = K8x::from(cf[i]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int c = cb * 32 + l32;
          const int o = min(c, C - 1) * P3 + C + min(q2, SC - 1) * E;
          K8x af = K8x::load(w3H + o, w3L + o);
          if (c >= C || q2 >= SC) af = K8x::zero();
          mma_step_x3(zc[cb], zl[cb], af, cx);
        }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) x3_fold(zc[cb], zl[cb]);
    } else {
#pragma unroll
    for (int s = 0; s < NSTM; ++s) {
      // B operand: registers 8*(s&1) ... +7 of block s >> 1  <->  channels 16 s + 8 (j >> 2) + 4 half + (j & 3)
      K8<T> bf;
#pragma unroll
      for (int j = 0; j < 8; ++j) bf.set(j, acc[s >> 1][8 * (s & 1) + j]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int c = cb * 32 + l32;
        const T *wr = w3s + min(c, C - 1) * P3 + 16 * s + 4 * half;
        K8<T> af;
        const K4<T> lo = K4<T>::load(wr), hi = K4<T>::load(wr + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          af.set(j, (c < C && 16 * s + 4 * half + j < C) ? lo.get(j) : 0.f);
          af.set(4 + j, (c < C && 16 * s + 8 + 4 * half + j < C) ? hi.get(j) : 0.f);
        }
        mma_step(zc[cb], af, bf);
      }
    }
#pragma unroll
    for (int i = 0; i < NSTC; ++i) {
      const int q2 = 2 * i + half;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int c = cb * 32 + l32;
        K8<T> af = K8<T>::load(w3s + min(c, C - 1) * P3 + C + min(q2, SC - 1) * E);
        if (c >= C || q2 >= SC) af = K8<T>::zero();
        mma_step(zc[cb], af, cf[i]);
      }
    }
    }
    // ---- epilogue: + bias + m (+ per-clip bias), store, GroupNorm partial of the stored values ------------------------------
    float xs[NCB][4][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c0 = cb * 32 + half * 4 + 8 * v;
        if (c0 >= C) continue;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(ep + 64 + c0);   // bias3 + per-clip add
        float val[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = zc[cb][4 * v + e] + bias[e] + acc[cb][4 * v + e];
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f<T>(val[e]);
        if (rvalid) {
          T *op = static_cast<T *>(a.out) + grow * C + c0;
          if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2 *>(op) = *reinterpret_cast<const uint2 *>(o);
          else *reinterpret_cast<f32x4 *>(op) = *reinterpret_cast<const f32x4 *>(o);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[cb][v][e] = rvalid ? to_f(o[e]) : 0.f;
      }
    if (a.stats_out) {
      const int vrows = min(32, rows - t * 32);
      if constexpr (C == 8) {
        const float cnt = (float)vrows;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m = div_t<FAST>(half_sum32(xs[0][0][e], half), cnt);
          const float d = rvalid ? xs[0][0][e] - m : 0.f;
          const float q = half_sum32(d * d, half);
          if (l32 == 0) {
            float *pp = part + ((size_t)t * a.G + half * 4 + e) * 2;
            pp[0] = m;
            pp[1] = q;
          }
        }
      } else {
        const float cnt = (float)vrows * (float)cpg;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float2 hs = half_sums((xs[cb][v][0] + xs[cb][v][1]) + (xs[cb][v][2] + xs[cb][v][3]));
            const float m = div_t<FAST>(cpg == 8 ? hs.x + hs.y : (half ? hs.y : hs.x), cnt);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = rvalid ? xs[cb][v][e] - m : 0.f;
              q = fmaf(d, d, q);
            }
            const float2 hq = half_sums(q);
            const float qq = cpg == 8 ? hq.x + hq.y : (half ? hq.y : hq.x);
            if (l32 == 0 && (cpg == 4 || half == 0)) {
              float *pp = part + ((size_t)t * a.G + (cb * 32 + half * 4 + 8 * v) / cpg) * 2;
              pp[0] = m;
              pp[1] = qq;
            }
          }
      }
    }
  }
  if (a.stats_out) {
    __syncthreads();
    for (int g = wave; g < a.G; g += NW) {
      const bool on = lane < ntile;
      const float *pp = part + ((size_t)(on ? lane : 0) * a.G + g) * 2;
      const float n_t = on ? (float)min(32, rows - lane * 32) * (float)cpg : 0.f;
      const float m_t = on ? pp[0] : 0.f, q_t = on ? pp[1] : 0.f;
      const float2 sn = half_sums(n_t), sm = half_sums(n_t * m_t);
      const float n = sn.x + sn.y, mean = div_t<FAST>(sm.x + sm.y, n);
      const float d = m_t - mean;
      const float2 sq2 = half_sums(fmaf(n_t * d, d, q_t));
      if (lane == 0) {
        float *so = a.stats_out + (((size_t)b * a.nchw + ch) * a.G + g) * 2;
        so[0] = mean;
        so[1] = sq2.x + sq2.y;
      }
    }
  }
}

template <typename T> size_t tail_lds_bytes(int C, int C2, int rw) {
  return (size_t)(4 * C + 128 + kThinMaxTiles * kThinMaxG * 2) * sizeof(float) +
         ((size_t)C * (3 * C + 8) + (size_t)C * (C + C2 + 8) + (size_t)(rw + 4) * (C + 8)) * sizeof(T);
}

template <typename T, int C, int C2, bool X3 = false> hipError_t tail_go(const ThinTailArgs &a, hipStream_t s) {
  const size_t lds = tail_lds_bytes<T>(C, C2, a.rw);
  auto kern = thin_tail_kernel<T, C, C2, X3>;
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      raised = true;
    }
  }
  int nw = (a.rw + 31) / 32;
  if (nw > thin_max_waves(C)) nw = thin_max_waves(C);
  if (X3 && nw > 8) nw = 8;
  if ((a.rw + 2) * (C / 8) > 6 * nw * 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3(a.B * a.nchw), dim3(nw * 64), lds, s, a);
  return hipGetLastError();
}

template <typename T> hipError_t tail_dispatch(const ThinTailArgs &a, hipStream_t s) {
  if (a.C == 8 && a.C2 == 8) return tail_go<T, 8, 8>(a, s);
  if (a.C == 32 && a.C2 == 32) return tail_go<T, 32, 32>(a, s);
  if (a.C == 64 && a.C2 == 32) return tail_go<T, 64, 32>(a, s);
  if (a.C == 64 && a.C2 == 64) return tail_go<T, 64, 64>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace

bool thin_tail_supported(int dt, const ThinTailArgs &a) {
  const bool shape = (a.C == 8 && a.C2 == 8) || (a.C == 32 && a.C2 == 32) || (a.C == 64 && (a.C2 == 32 || a.C2 == 64));
  if (!shape || a.rw < 32 || a.rw % 32 || a.rw > 32 * kThinMaxTiles) return false;
  if (a.G < 1 || a.G > kThinMaxG || a.C % a.G) return false;
  const int cpg = a.C / a.G;
  if (a.C == 8 ? cpg != 1 : (cpg != 4 && cpg != 8)) return false;
  if (!a.ss || !a.stats_in) return false;
  {
    int nw = (a.rw + 31) / 32;
    if (nw > thin_max_waves(a.C)) nw = thin_max_waves(a.C);
    if ((a.rw + 2) * (a.C / 8) > 6 * nw * 64) return false;
  }
  const size_t lds = dt == F32 ? tail_lds_bytes<float>(a.C, a.C2, a.rw) : tail_lds_bytes<bf16>(a.C, a.C2, a.rw);
  return lds <= 160 * 1024;
}

hipError_t launch_thin_tail(int dt, const ThinTailArgs &a, hipStream_t s) {
  if (!thin_tail_supported(dt, a)) return hipErrorInvalidValue;
  if (d0_tail_supported(a)) return launch_d0_tail(dt, a, s);
  if (dt == F32 && a.x3) {   // split-operand instantiations (the 8-channel level stays on the vector / fp32 kernels)
    if (a.C == 32 && a.C2 == 32) return tail_go<float, 32, 32, true>(a, s);
    if (a.C == 64 && a.C2 == 32) return tail_go<float, 64, 32, true>(a, s);
    if (a.C == 64 && a.C2 == 64) return tail_go<float, 64, 64, true>(a, s);
  }
  return SF_DISPATCH_T(dt, tail_dispatch<T>(a, s));
}

ThinPlan conv_thin_plan(int B, int L, int C) {
  ThinPlan p;
  static const int rows8 = [] {   // tuning hook: positions per workgroup on the 8-channel level
    const char *e = tune_env("SF_THIN_ROWS8");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 0;
  }();
  // a wave's fixed cost (weights, prologue table, epilogue bookkeeping) is amortised over 32 positions x C channels
  // per tile: the 8-channel level gives each wave several tiles
  static const int rows_cap = [] {   // tuning hook: upper bound of positions per workgroup above 8 channels
    const char *e = tune_env("SF_THIN_MAXROWS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 352;   // 64 channels: 352 positions x 8 waves (two workgroups per CU; 6 staged vectors per thread bound the chunk);
                               // one workgroup of 16 waves x 1024 positions measured 1.4 % slower on configs[2] and the 2^18-sample shape
                               // (profiles/r3_k_ab_thin_waves.txt).  (512 until round 3: with two branches at the guidance batch 1024 / 2048 measured +3.6 % on configs[2], +3 % at batch 32, +4.4 % on the 2^18-sample shape, profiles/r3_g_ab_thin_rows.txt)
  }();
  // 8-channel level: 992 = 16 x 62 where the vector kernels run (conv_d0.hip) -- whole passes for them, 31 MFMA tiles for the up
  // convolution that shares the chunking; against 2048 it measured +1.6 % on 32 evaluations per step and no change on 64
  // (profiles/r3_j_ab_d0.txt).  The MFMA formulation prefers the long chunk.
  static const int rows32 = [] {   // tuning hook: the same bound on the 32-channel level alone
    const char *e = tune_env("SF_THIN_MAXROWS32");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 512;
  }();
  const int max_rows = C <= 8 ? (rows8 > 0 ? rows8 : (d0_enabled(B, L) ? 992 : 2048)) : (C == 32 && rows32 > 0 ? rows32 : rows_cap);
  static const int wgs = [] {   // tuning hook: workgroups a launch aims for
    const char *e = tune_env("SF_THIN_WGS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 256;
  }();
  long target = ((long)B * L + wgs - 1) / wgs;   // positions per workgroup
  int rw = (int)((target + 31) / 32) * 32;
  if (rw < 32) rw = 32;
  if (rw > max_rows) rw = max_rows;
  p.rw = rw;
  p.nchw = (L + rw - 1) / rw;
  return p;
}

bool conv_thin_supported(int dt, const ConvThinArgs &a) {
  bool shape = false;
  for (const ThinShape &t : kThinShapes) shape = shape || (a.C == t.cin && a.taps == t.taps && a.C2 == t.c2 && a.pro == t.pro && a.N == t.nout);
  if (!shape) return false;
  if (a.rw < 32 || a.rw % 32 || a.rw > 32 * kThinMaxTiles) return false;
  if (a.up_shift < 0 || a.up_shift > 4 || (a.Ls << a.up_shift) != a.L) return false;
  if (a.up_shift && (a.C2 || a.pro || a.res_self)) return false;
  if (a.pro == 1 || a.stats_out) {   // GroupNorm bookkeeping: a lane's four channels must sit in one group (or be four groups)
    if (a.G < 1 || a.G > kThinMaxG || a.N % a.G) return false;
    const int cpg = a.N / a.G;
    if (a.N == 8 ? cpg != 1 : (cpg != 4 && cpg != 8)) return false;
  }
  {   // staging registers: the source rows of a workgroup must fit 6 vectors per thread (see thin_go)
    const int ncb = (a.N + 31) / 32;
    int nw = ((a.rw + 31) / 32) * ncb;
    if (nw > thin_max_waves(a.N)) nw = thin_max_waves(a.N);
    nw = (nw / ncb) * ncb;
    if (((a.rw >> a.up_shift) + 3) * (a.C / 8) > 6 * nw * 64) return false;
  }
  const size_t lds = dt == F32 ? thin_lds_bytes<float>(a.C, a.taps, a.rw, a.up_shift) : thin_lds_bytes<bf16>(a.C, a.taps, a.rw, a.up_shift);
  return lds <= 160 * 1024;
}

hipError_t launch_conv_thin(int dt, const ConvThinArgs &a, hipStream_t s) {
  if (!conv_thin_supported(dt, a)) return hipErrorInvalidValue;
  if (d0_conv_supported(a)) return launch_d0_conv(dt, a, s);
  if (dt == F32 && a.x3) {
    const hipError_t e = thin_dispatch_x3(a, s);
    if (e != hipErrorInvalidValue) return e;   // (shapes without a split instantiation run in fp32)
  }
  return SF_DISPATCH_T(dt, thin_dispatch<T>(a, s));
}

}  // namespace sf
