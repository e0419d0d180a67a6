// Shared device helpers for the gfx950 kernels (wave64, MFMA, 16-byte vector access).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"   // X3_F16 / X3_BF16

namespace sf {

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int WAVE = 64;

// 16-byte vector of T: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <> struct Vec16<f16> {
  static constexpr int N = 8;
  f16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (f16)x; }
};

// 16-bit MFMA operand fragment of T (8 consecutive k of one row) and the 32x32x16 product on it
template <typename T> struct Frag16 {
  using type = bf16x8;
};
template <> struct Frag16<f16> {
  using type = f16x8;
};
__device__ __forceinline__ f32x16 mfma32x16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32x16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// ---- fp32-accurate products from split 16-bit operands ("x3", the parity-grade fast path) ------------------------------------------------
// An fp32 operand is split in registers into two 16-bit parts, a = hi + lo / SCALE, and a product is three 16-bit MFMAs into two fp32
// accumulators:  accM += hi_a hi_b;  accL += hi_a lo_b + lo_a hi_b;  result = accM + accL / SCALE  (the lo lo term is dropped).
//   mode 1 (forward passes): hi = fp16(a), lo = fp16((a - hi) * 2048) -- 22 significant bits per operand, the scale keeps lo out of the
//     fp16 subnormal range.  Measured against fp64 on a 256 x 256 x 3072 GEMM: rel-L2 7.5e-8 (plain fp32 MFMA 3.5e-7, one fp16 product
//     2.9e-4).  Range: |a| < 65504 (fp16 maximum), as for the fp16 engine; values below ~1e-7 of the tensor's typical magnitude lose bits.
//   mode 2 (backward passes, whose gradients span the whole fp32 exponent range): hi = bf16(a), lo = bf16(a - hi), SCALE = 1 -- 16 significant
//     bits per operand, 4.4e-6 on the same GEMM, no range restriction.
template <int MODE> struct X3P;
template <> struct X3P<X3_F16> {
  using v8 = f16x8;
  using elem = f16;
  static constexpr float SCALE = 2048.0f, INV = 1.0f / 2048.0f;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct X3P<X3_BF16> {
  using v8 = bf16x8;
  using elem = bf16;
  static constexpr float SCALE = 1.0f, INV = 1.0f;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <int MODE> __device__ __forceinline__ void x3_split1(float v, typename X3P<MODE>::elem &hi, typename X3P<MODE>::elem &lo) {
  using E = typename X3P<MODE>::elem;
  const E h = (E)v;
  hi = h;
  lo = (E)((v - (float)h) * X3P<MODE>::SCALE);
}
template <int MODE> __device__ __forceinline__ void x3_split(const f32x4 p, const f32x4 q, typename X3P<MODE>::v8 &hi, typename X3P<MODE>::v8 &lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    typename X3P<MODE>::elem h, l;
    x3_split1<MODE>(p[e], h, l);
    hi[e] = h;
    lo[e] = l;
    x3_split1<MODE>(q[e], h, l);
    hi[4 + e] = h;
    lo[4 + e] = l;
  }
}
// the bf16 split needs no scale, so its three products can share ONE accumulator (half the accumulator registers: a second workgroup per CU)
__device__ __forceinline__ void x3_mfma1_bf16(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 &acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);   // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}
// four consecutive channels c0 .. c0 + 3 (c0 % 4 == 0) of an activation row in the pre-split layout [C / 32][hi 32 | lo' 32] fp16 (mode 1):
// what ConvGemmArgs::src_x3 reads.  `row` points at the row's first byte (the row is C * 4 bytes long, as in fp32).
__device__ __forceinline__ void st4_x3(void *row, int c0, const f32x4 v) {
  typedef f16 f16x4_s __attribute__((ext_vector_type(4)));
  f16x4_s h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f16 a, b;
    x3_split1<X3_F16>(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  unsigned char *p = static_cast<unsigned char *>(row) + (c0 >> 5) * 128 + (c0 & 31) * 2;
  *reinterpret_cast<f16x4_s *>(p) = h;
  *reinterpret_cast<f16x4_s *>(p + 64) = l;
}
// the three products of one 32x32x16 step
template <int MODE>
__device__ __forceinline__ void x3_mfma(const typename X3P<MODE>::v8 ah, const typename X3P<MODE>::v8 al, const typename X3P<MODE>::v8 bh,
                                        const typename X3P<MODE>::v8 bl, f32x16 &accM, f32x16 &accL) {
  accM = X3P<MODE>::mfma(ah, bh, accM);
  accL = X3P<MODE>::mfma(ah, bl, accL);
  accL = X3P<MODE>::mfma(al, bh, accL);
}

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T *p) {
  Vec16<T> r;
  u32x4 raw = *reinterpret_cast<const u32x4 *>(p);
  r.v = __builtin_bit_cast(decltype(r.v), raw);
  return r;
}
template <typename T> __device__ __forceinline__ void st16(T *p, const Vec16<T> &x) {
  *reinterpret_cast<u32x4 *>(p) = __builtin_bit_cast(u32x4, x.v);
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
  Vec16<T> r;
  u32x4 z = {0u, 0u, 0u, 0u};
  r.v = __builtin_bit_cast(decltype(r.v), z);
  return r;
}

template <typename T> __device__ __forceinline__ float to_f(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

// Body of a hosted-prefetch workgroup (kernels.h, Prefetch): workgroup `w` of `pf.wgs` reads its slice of the range, all loads of a
// thread in flight together, nothing kept.  The caller returns right after it.
template <typename P> __device__ __forceinline__ void prefetch_slice(const P &pf, int w, int nthreads) {
  const uint4 *base = static_cast<const uint4 *>(pf.ptr);
  const unsigned n = pf.bytes >> 4, stride = (unsigned)pf.wgs * (unsigned)nthreads;
  unsigned acc = 0;
  for (unsigned i = (unsigned)w * (unsigned)nthreads + threadIdx.x; i < n; i += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = base[min(i + k * stride, n - 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc ^= v[k].x;
  }
  asm volatile("" ::"v"(acc));   // keeps the loads alive without a store
}

// exp: accurate on the fp32 parity path, v_exp_f32 based on the bf16 path.
template <bool FAST> __device__ __forceinline__ float exp_t(float x) {
  if constexpr (FAST) return __expf(x);
  else return expf(x);
}
// 16-bit paths: x * rcp(1 + exp(-x)) -- v_exp_f32 + v_rcp_f32 (1 ulp each, far inside a 16-bit result) instead of the IEEE
// division sequence (v_div_scale / v_rcp / 4 v_fma / v_div_fmas / v_div_fixup: ~10 VALU instructions per element, which made the
// GroupNorm+SiLU prologues of the thin-level kernels VALU-bound: SQ_ACTIVE_INST_VALU = 73 % of their issue cycles at batch 64)
template <bool FAST> __device__ __forceinline__ float silu_t(float x) {
  if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  else return x / (1.0f + expf(-x));
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// epilogue activation codes shared by the conv kernels: 0 none, 1 relu, 2 gelu(erf), 3 silu(gelu(v))
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return gelu_erf(v);
  if (act == 3) {
    float g = gelu_erf(v);
    return g / (1.0f + expf(-g));
  }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Wave-wide sum with DPP row operations (VALU only -- __shfl_xor is ds_bpermute, i.e. LDS crossbar traffic); every lane
// receives the total.  Fixed order: deterministic.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_mov_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_mov_f<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x124, 0xf>(v);   // row_ror:4
  v += dpp_mov_f<0x128, 0xf>(v);   // row_ror:8      -> every lane holds the sum of its row of 16
  v += dpp_mov_f<0x142, 0xa>(v);   // row_bcast:15   -> rows 1 and 3 add the total of the row before
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)) +
         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum over aligned groups of 8 consecutive lanes (every lane of the group receives it): two quad permutes + half-row mirror
__device__ __forceinline__ float sum8_dpp(float v) {
  v += dpp_mov_f<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f<0x141, 0xf>(v);   // row_half_mirror: lane i <-> 7 - i inside each 8
  return v;
}

// Chan/Welford merge of (n, mean, M2) partials.
__device__ __forceinline__ void welford_merge(float &n, float &mean, float &m2, float nb, float mb, float m2b) {
  if (nb <= 0.f) return;
  float tot = n + nb;
  float delta = mb - mean;
  mean += delta * (nb / tot);
  m2 += m2b + delta * delta * (n * nb / tot);
  n = tot;
}

// The same with v_rcp_f32 (1 ulp) instead of two IEEE divisions (~10 VALU instructions each): the 16-bit engines' thin-level
// kernels run ~18 divisions per wave in their statistics bookkeeping, 12 % of their instruction stream.
template <bool FAST> __device__ __forceinline__ float div_t(float a, float b) {
  if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
  else return a / b;
}
template <bool FAST> __device__ __forceinline__ void welford_merge_t(float &n, float &mean, float &m2, float nb, float mb, float m2b) {
  if constexpr (!FAST) {
    welford_merge(n, mean, m2, nb, mb, m2b);
  } else {
    if (nb <= 0.f) return;
    const float tot = n + nb, r = __builtin_amdgcn_rcpf(tot);
    const float delta = mb - mean;
    mean += delta * (nb * r);
    m2 += m2b + delta * delta * (n * nb * r);
    n = tot;
  }
}

// Final GroupNorm statistics of one (clip, group) from its <= 32 chunk partials (mean, M2): each of the 32 lanes of
// a half-wave loads one chunk (all loads in flight together), then a fixed-shape shuffle tree merges them (Chan) --
// deterministic, one memory latency.  Every lane of the half-wave must call it; every lane gets the result.
__device__ __forceinline__ float2 gn_merge32(const float *__restrict__ sl, int G, int nch, int chunk_rows, int L, int cpg,
                                             float eps, int lane32) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  if (lane32 < nch) {
    const int rows = min(chunk_rows, L - lane32 * chunk_rows);
    n = (float)rows * (float)cpg;
    mean = sl[(size_t)lane32 * G * 2];
    m2 = sl[(size_t)lane32 * G * 2 + 1];
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const float nb = __shfl_down(n, off, 32), mb = __shfl_down(mean, off, 32), qb = __shfl_down(m2, off, 32);
    welford_merge(n, mean, m2, nb, mb, qb);
  }
  const float mu = __shfl(mean, 0, 32), var = __shfl(m2, 0, 32) / __shfl(n, 0, 32);
  return make_float2(mu, rsqrtf(var + eps));
}

}  // namespace sf
