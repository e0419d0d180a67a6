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
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int kThinMaxTiles = 16;   // 32-position tiles per workgroup (RW <= 512)
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
__device__ __forceinline__ void mma_step(f32x16 &acc, const K8<float> &a, const K8<float> &b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.get(j), b.get(j), acc, 0, 0, 0);
}

// (mean, rstd) of one (clip, group) from its nch chunk partials, any nch: lane j of the half-wave folds partials
// j, j+32, ... in order, then the fixed shuffle tree of gn_merge32 -- deterministic.
__device__ __forceinline__ float2 gn_merge_n(const float *__restrict__ sl, int G, int nch, int chunk_rows, int L, int cpg, float eps,
                                             int lane32) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int i = lane32; i < nch; i += 32) {
    const int rows = min(chunk_rows, L - i * chunk_rows);
    welford_merge(n, mean, m2, (float)rows * (float)cpg, sl[(size_t)i * G * 2], sl[(size_t)i * G * 2 + 1]);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const float nb = __shfl_down(n, off, 32), mb = __shfl_down(mean, off, 32), qb = __shfl_down(m2, off, 32);
    welford_merge(n, mean, m2, nb, mb, qb);
  }
  const float mu = __shfl(mean, 0, 32), var = __shfl(m2, 0, 32) / __shfl(n, 0, 32);
  return make_float2(mu, rsqrtf(var + eps));
}

template <typename T, int C, int TAPS, int C2, int PRO>
__global__ __launch_bounds__(1024) void conv_thin_kernel(const ConvThinArgs a) {
  constexpr int E = 8;
  constexpr int QC = C / E;                 // channel octets of the first source
  constexpr int S1 = TAPS * QC, S2 = C2 / E, S = S1 + S2, NSTEP = S / 2;
  static_assert(S % 2 == 0, "slot count must be even");
  constexpr int NCB = C / 32;               // 32-wide blocks of output channels
  constexpr int SROW = C + E;               // LDS row pitch in elements (16-byte skew against bank conflicts)
  constexpr int HALO = TAPS / 2;
  constexpr bool FAST = !std::is_same<T, float>::value;
  constexpr bool KEEP_W = sizeof(T) == 2;   // bf16: the wave's weight fragments stay in registers

  extern __shared__ __align__(16) unsigned char smem[];
  float *sc = reinterpret_cast<float *>(smem);
  float *sh = sc + C;
  float *part = sh + C;   // [tile][G][3]
  T *tile = reinterpret_cast<T *>(part + kThinMaxTiles * kThinMaxG * 3);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.x / a.nchw, ch = blockIdx.x - b * a.nchw;
  const int r0 = ch * a.rw;
  const int rows = min(a.rw, a.L - r0);
  const int ntile = (rows + 31) >> 5;
  const int K = S * E;
  const T *src = static_cast<const T *>(a.src);
  const T *wgt = static_cast<const T *>(a.w);
  const int cb = wave % NCB;   // launch guarantees NW % NCB == 0: a wave keeps one block of output channels

  // ---- weights of this wave (issued first: they do not depend on anything) ----------------------------------
  K8<T> wf[KEEP_W ? NSTEP : 1];
  const T *wrow = wgt + (size_t)(cb * 32 + l32) * K + half * E;
  if constexpr (KEEP_W) {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wf[s] = K8<T>::load(wrow + s * 2 * E);
  }

  // ---- prologue parameters -> LDS ----------------------------------------------------------------------------
  if constexpr (PRO == 1) {
    const int cpg = C / a.G;
    for (int g = tid >> 5; g < a.G; g += blockDim.x >> 5) {
      const float2 st = gn_merge_n(a.stats_in + ((size_t)b * a.nch_in * a.G + g) * 2, a.G, a.nch_in, a.chunk_in, a.L, cpg, a.eps, l32);
      if (l32 < cpg) {
        const int c = g * cpg + l32;
        const float s = st.y * a.gamma[c];
        sc[c] = s;
        sh[c] = a.beta[c] - st.x * s;
      }
    }
    __syncthreads();
  } else if constexpr (PRO == 2) {
    if (tid < C) {
      sc[tid] = a.ss ? 1.0f + a.ss[(size_t)b * a.ss_ld + tid] : 1.0f;
      sh[tid] = a.ss ? a.ss[(size_t)b * a.ss_ld + C + tid] : 0.0f;
    }
    __syncthreads();
  }

  // ---- stage rows [r0 - HALO, r0 + rows + HALO) with the prologue applied once --------------------------------
  {
    const int total = (rows + 2 * HALO) * QC;
    for (int idx = tid; idx < total; idx += blockDim.x) {
      const int rr = idx / QC, q = idx - rr * QC;
      const int pos = r0 - HALO + rr;
      const bool ok = pos >= 0 && pos < a.L;
      K8<T> v = ok ? K8<T>::load(src + ((size_t)b * a.L + pos) * a.src_ld + q * E) : K8<T>::zero();
      if constexpr (PRO == 1) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const float y = fmaf(v.get(j), sc[q * E + j], sh[q * E + j]);
          v.set(j, ok ? silu_t<FAST>(y) : 0.f);
        }
      } else if constexpr (PRO == 2) {
        // LayerNorm over the C channels of the row: its QC octets sit on QC consecutive lanes
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < E; ++j) sum += v.get(j);
#pragma unroll
        for (int o = QC >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const float d = v.get(j) - mean;
          sq = fmaf(d, d, sq);
        }
#pragma unroll
        for (int o = QC >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq / (float)C + a.eps);
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const float y = (v.get(j) - mean) * rstd;
          v.set(j, fmaf(y, sc[q * E + j], sh[q * E + j]));
        }
      }
      v.store(tile + rr * SROW + q * E);
    }
  }
  __syncthreads();

  // ---- tiles ---------------------------------------------------------------------------------------------------
  const int cpg = C / a.G;
  for (int item = wave; item < ntile * NCB; item += NW) {
    const int t = item / NCB;
    const int row_l = t * 32 + l32;
    const bool rvalid = row_l < rows;
    const size_t grow = (size_t)b * a.L + r0 + (rvalid ? row_l : 0);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      K8<T> bf;
      const int slot = 2 * s + half;   // S1 is even: both halves of a step read the same source
      if (2 * s + 1 < S1) {   // both halves read the staged first source
        const int tap = slot / QC, q = slot - tap * QC;
        bf = K8<T>::load(tile + (row_l + tap) * SROW + q * E);
      } else if (2 * s >= S1) {   // both halves read the second source from global memory
        const int q2 = slot - S1;
        bf = rvalid ? K8<T>::load(static_cast<const T *>(a.src2) + grow * a.src2_ld + q2 * E) : K8<T>::zero();
      } else {   // S1 odd: cannot happen for the instantiated shapes (S1 even)
        bf = K8<T>::zero();
      }
      if constexpr (KEEP_W) mma_step(acc, wf[s], bf);
      else mma_step(acc, K8<T>::load(wrow + s * 2 * E), bf);
    }

    // ---- epilogue: lane = position row_l, registers 4v..4v+3 = channels cb*32 + half*4 + 8v + {0..3} -----------
    float pn[4], pm[4], pq[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c0 = cb * 32 + half * 4 + 8 * v;
      float val[4];
      const f32x4 bias = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = acc[4 * v + e] + bias[e];
      if (a.res_self) {   // residual = the staged (modulated) input itself
        const T *rp = tile + (row_l + HALO) * SROW + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] += to_f(rp[e]);
      } else if (a.res && rvalid) {
        const T *rp = static_cast<const T *>(a.res) + grow * a.res_ld + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] += to_f(rp[e]);
      }
      if (a.badd) {
        const f32x4 ba = *reinterpret_cast<const f32x4 *>(a.badd + (size_t)b * a.badd_ld + c0);
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
      // GroupNorm partial of the STORED values of this lane's four channels (one group: cpg is 4 or 8)
      if (a.stats_out) {
        const float x0 = to_f(o[0]), x1 = to_f(o[1]), x2 = to_f(o[2]), x3 = to_f(o[3]);
        const float m = 0.25f * ((x0 + x1) + (x2 + x3));
        const float d0 = x0 - m, d1 = x1 - m, d2 = x2 - m, d3 = x3 - m;
        pn[v] = rvalid ? 4.f : 0.f;
        pm[v] = rvalid ? m : 0.f;
        pq[v] = rvalid ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
      }
    }
    if (a.stats_out) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
          const float nb = __shfl_down(pn[v], off, 32), mb = __shfl_down(pm[v], off, 32), qb = __shfl_down(pq[v], off, 32);
          welford_merge(pn[v], pm[v], pq[v], nb, mb, qb);
        }
        if (cpg == 8) {   // the two half-waves hold the two halves of the same group
          const float nb = __shfl(pn[v], 32, 64), mb = __shfl(pm[v], 32, 64), qb = __shfl(pq[v], 32, 64);
          if (lane == 0) welford_merge(pn[v], pm[v], pq[v], nb, mb, qb);
        }
        const int g = (cb * 32 + half * 4 + 8 * v) / cpg;
        if (l32 == 0 && (cpg == 4 || half == 0)) {
          float *pp = part + ((size_t)t * a.G + g) * 3;
          pp[0] = pn[v];
          pp[1] = pm[v];
          pp[2] = pq[v];
        }
      }
    }
  }

  // ---- (mean, M2) of this workgroup's output per group, tiles folded in order ------------------------------------
  if (a.stats_out) {
    __syncthreads();
    if (tid < a.G) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
      for (int t = 0; t < ntile; ++t) {
        const float *pp = part + ((size_t)t * a.G + tid) * 3;
        welford_merge(n, mean, m2, pp[0], pp[1], pp[2]);
      }
      float *so = a.stats_out + (((size_t)b * a.nchw + ch) * a.G + tid) * 2;
      so[0] = mean;
      so[1] = m2;
    }
  }
}

template <typename T> size_t thin_lds_bytes(int C, int taps, int rw) {
  const int halo = taps / 2;
  const size_t rows = (size_t)((rw + 31) / 32) * 32 + 2 * halo + 1;
  return (size_t)(2 * C + kThinMaxTiles * kThinMaxG * 3) * sizeof(float) + rows * (C + 8) * sizeof(T);
}

template <typename T, int C, int TAPS, int C2, int PRO> hipError_t thin_go(const ConvThinArgs &a, hipStream_t s) {
  const size_t lds = thin_lds_bytes<T>(C, TAPS, a.rw);
  auto kern = conv_thin_kernel<T, C, TAPS, C2, PRO>;
  if (lds > 64 * 1024) {
    static bool raised = false;   // per instantiation
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      raised = true;
    }
  }
  constexpr int NCB = C / 32;
  int nw = ((a.rw + 31) / 32) * NCB;
  if (nw > 16) nw = 16;
  nw = (nw / NCB) * NCB;
  hipLaunchKernelGGL(kern, dim3(a.B * a.nchw), dim3(nw * 64), lds, s, a);
  return hipGetLastError();
}

template <typename T> hipError_t thin_dispatch(const ConvThinArgs &a, hipStream_t s) {
#define SF_THIN(CC, TT, C22, PP) \
  if (a.C == CC && a.taps == TT && a.C2 == C22 && a.pro == PP) return thin_go<T, CC, TT, C22, PP>(a, s)
  SF_THIN(32, 3, 0, 1);
  SF_THIN(64, 3, 0, 1);
  SF_THIN(32, 1, 32, 2);
  SF_THIN(64, 1, 32, 2);
  SF_THIN(64, 1, 64, 2);
#undef SF_THIN
  return hipErrorInvalidValue;
}

}  // namespace

ThinPlan conv_thin_plan(int B, int L) {
  ThinPlan p;
  long target = ((long)B * L + 255) / 256;   // positions per workgroup for ~256 workgroups
  int rw = (int)((target + 31) / 32) * 32;
  if (rw < 32) rw = 32;
  if (rw > 32 * kThinMaxTiles) rw = 32 * kThinMaxTiles;
  p.rw = rw;
  p.nchw = (L + rw - 1) / rw;
  return p;
}

bool conv_thin_supported(int dt, const ConvThinArgs &a) {
  if (a.G < 1 || a.G > kThinMaxG || a.C % a.G) return false;
  const int cpg = a.C / a.G;
  if (cpg != 4 && cpg != 8) return false;
  if (a.rw < 32 || a.rw % 32 || a.rw > 32 * kThinMaxTiles) return false;
  const bool shape = (a.taps == 3 && a.C2 == 0 && a.pro == 1 && (a.C == 32 || a.C == 64)) ||
                     (a.taps == 1 && a.pro == 2 && ((a.C == 32 && a.C2 == 32) || (a.C == 64 && (a.C2 == 32 || a.C2 == 64))));
  if (!shape) return false;
  const size_t lds = dt == F32 ? thin_lds_bytes<float>(a.C, a.taps, a.rw) : thin_lds_bytes<bf16>(a.C, a.taps, a.rw);
  return lds <= 160 * 1024;
}

hipError_t launch_conv_thin(int dt, const ConvThinArgs &a, hipStream_t s) {
  if (!conv_thin_supported(dt, a)) return hipErrorInvalidValue;
  return dt == F32 ? thin_dispatch<float>(a, s) : thin_dispatch<bf16>(a, s);
}

}  // namespace sf
