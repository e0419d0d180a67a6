// Lean variant of the wave-split-K implicit GEMM (conv_gemm_sk.hip) for the shapes that dominate the U-Net:
// 1-D convolutions with <= 3 taps whose channel count is a multiple of the 128-wide K chunk (C = 128..1024),
// optionally followed by a concatenated second source (InjectChannels).  Everything that made the generic
// kernel issue-bound at one wave per SIMD is hoisted out of the K loop:
//   * per-row source offsets for every tap are computed ONCE (32-bit byte offsets);
//   * the tap of a chunk is wave-uniform (scalar), so picking the row offset is two selects;
//   * loads are `buffer_load_dwordx4` through wave-uniform buffer descriptors: rows outside the padding, rows
//     beyond M / N and K tails get an out-of-range offset and the hardware returns zeros -- no predication,
//     no zero-fill pass, no 64-bit address arithmetic in the loop.
// The K loop is then: 2 scalar ops + (1 select + 1 add + 1 load) per staged vector, LDS write, barrier, MFMAs.
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr unsigned OOB = 0x80000000u;   // beyond every buffer: the load returns zero

template <typename T> __device__ __forceinline__ Vec16<T> buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  Vec16<T> v;
  u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  v.v = __builtin_bit_cast(decltype(v.v), raw);
  return v;
}

// KW = K elements one wave multiplies per iteration (32 or 64); the workgroup stages BKT = 4*KW per iteration.
// NSET = K chunks in flight per workgroup.  The weights of a denoising step stream from HBM (they do not fit the
// Infinity Cache), so a workgroup needs latency x bandwidth bytes outstanding: 2 chunks (32 KB) cap a CU at ~45 GB/s.
template <typename T, int BM, int BN, bool CAT, int KW, int NSET>
__global__ __launch_bounds__(256) void conv_gemm_fast_kernel(const ConvGemmArgs a, const int mtiles, const int ntiles, const int swz,
                                                             const unsigned bytesA, const unsigned bytesA2, const unsigned bytesW) {
  constexpr int BKT = 4 * KW;
  constexpr int VEC = Vec16<T>::N;
  constexpr int ES = (int)sizeof(T);
  constexpr int VPR = BKT / VEC;
  constexpr int RPP = 256 / VPR;
  constexpr int PA = BM / RPP, PB = BN / RPP;
  constexpr int LD = BKT + 16 / ES;
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int LDR = BN + 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *As = reinterpret_cast<T *>(smem);
  T *Bs = As + BM * LD;
  float *red = reinterpret_cast<float *>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x, mt, nt;
  if (swz) {
    const int xcd = bid & 7, j = bid >> 3;
    nt = xcd + 8 * (j / mtiles);
    mt = j % mtiles;
  } else {
    nt = bid / mtiles;
    mt = bid % mtiles;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int srow = tid / VPR, svec = tid % VPR;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(CAT ? a.src2 : a.src), 0, CAT ? bytesA2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.w), 0, bytesW, 0x00020000);

  // ---- per-row byte offsets, once ---------------------------------------------------------------
  unsigned offA[PA][3], offA2[PA], offW[PB];
  const int pmax = (a.Lsrc << a.up_shift) - 1;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + i * RPP + srow;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int b = mm / a.Lout, l = mm - b * a.Lout;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int p = l * a.stride + t - a.pad;
      const bool ok = vm && t < a.taps && p >= 0 && p <= pmax;
      offA[i][t] = ok ? (unsigned)(((b * a.Lsrc + (p >> a.up_shift)) * a.src_ld + svec * VEC) * ES) : OOB;
    }
    offA2[i] = (CAT && vm) ? (unsigned)((m * a.src2_ld + svec * VEC) * ES) : OOB;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = n0 + i * RPP + srow;
    offW[i] = n < a.N ? (unsigned)((n * a.K + svec * VEC) * ES) : OOB;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct RegSet {
    Vec16<T> ra[PA], rb[PB];
  };
  RegSet rs[NSET];
  const int nkt = (a.K + BKT - 1) / BKT;

  // Streaming state of the gather, advanced once per prefetched chunk (chunks are requested in K order):
  //   cur[i]  byte offset of staged row i for the CURRENT tap (or of the second source once the taps are done)
  //   cb      byte offset of the chunk inside that tap's channels;  kb = byte offset of the chunk inside a W row
  unsigned cur[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) cur[i] = offA[i][0];
  unsigned cb = 0, kb = 0;
  int tap = 0;
  bool second = false;
  const unsigned tap_bytes = (unsigned)(a.cin * ES), kbytes = (unsigned)(a.K * ES), lane_kb = (unsigned)(svec * VEC * ES);

  auto prefetch = [&](RegSet &R) {
    // per lane, only in the last chunk of a ragged K: OR-ing the top bit makes the offset out of range (-> zeros)
    // without giving the compiler a select it could turn into a branch around the load
    const unsigned tmask = (kb + lane_kb >= kbytes) ? OOB : 0u;
#pragma unroll
    for (int i = 0; i < PB; ++i) R.rb[i] = buf_ld16<T>(rW, (offW[i] + kb) | tmask);
    const __amdgpu_buffer_rsrc_t rs = (CAT && second) ? rA2 : rA;
#pragma unroll
    for (int i = 0; i < PA; ++i) R.ra[i] = buf_ld16<T>(rs, (cur[i] + cb) | tmask);
    // advance (wave-uniform control flow; no memory operation inside)
    kb += (unsigned)(BKT * ES);
    cb += (unsigned)(BKT * ES);
    if (!second && cb >= tap_bytes) {
      cb = 0;
      ++tap;
      if (tap < a.taps) {
#pragma unroll
        for (int i = 0; i < PA; ++i) cur[i] = (tap == 1) ? offA[i][1] : offA[i][2];
      } else {
        second = true;
#pragma unroll
        for (int i = 0; i < PA; ++i) cur[i] = offA2[i];
      }
    }
  };
  auto stage = [&](RegSet &R) {
#pragma unroll
    for (int i = 0; i < PB; ++i) st16<T>(Bs + (i * RPP + srow) * LD + svec * VEC, R.rb[i]);
#pragma unroll
    for (int i = 0; i < PA; ++i) st16<T>(As + (i * RPP + srow) * LD + svec * VEC, R.ra[i]);
  };

  const int fr = lane & 31, fh = lane >> 5;
  const int kw0 = KW * wave;
  auto compute = [&]() {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < KW / 16; ++s) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(As + (i * 32 + fr) * LD + kw0 + 16 * s + 8 * fh);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(Bs + (j * 32 + fr) * LD + kw0 + 16 * s + 8 * fh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < KW / 8; ++q) {
        f32x4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(As + (i * 32 + fr) * LD + kw0 + (KW / 2) * fh + 4 * q);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const f32x4 *>(Bs + (j * 32 + fr) * LD + kw0 + (KW / 2) * fh + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- epilogue operands (bias, residual, per-clip scale / add): they depend on nothing the K loop computes, so with
  // one epilogue pass per thread (32x32 tiles) their loads are issued HERE and overlap the whole reduction instead of
  // costing a memory round trip after it ----------------------------------------------------------------------------
  const T *res = static_cast<const T *>(a.res);
  const bool has_res = res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
  constexpr int QN = BN / 4;
  constexpr int EIT = (BM * QN + 255) / 256;
  constexpr bool HOIST = EIT == 1;
  struct EpiOps {
    float bi[4], rv[4], sv[4], av[4];
  };
  auto epi_load = [&](int it) {
    EpiOps o;
    const int idx = tid + it * 256;
    const int ml = idx / QN, nq = idx - ml * QN;
    const int m = m0 + ml, nb = n0 + nq * 4;
    const int mc = min(m, a.M - 1);
    const int b = (has_bs || has_ba) ? mc / a.Lout : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int nc = min(nb + e, a.N - 1);
      o.bi[e] = a.bias ? a.bias[nc] : 0.f;
      o.rv[e] = has_res ? to_f(res[(size_t)mc * a.res_ld + nc]) : 0.f;
      o.sv[e] = has_bs ? a.bscale[(size_t)b * a.bscale_ld + nc] : 1.f;
      o.av[e] = has_ba ? a.badd[(size_t)b * a.badd_ld + nc] : 0.f;
    }
    return o;
  };
  EpiOps eo0;
  if constexpr (HOIST) eo0 = epi_load(0);

#pragma unroll
  for (int j = 0; j < NSET; ++j)
    if (j < nkt) prefetch(rs[j]);
  for (int kt = 0; kt < nkt; kt += NSET) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) {
      if (kt + j < nkt) {
        stage(rs[j]);
        __syncthreads();
        if (kt + j + NSET < nkt) prefetch(rs[j]);
        compute();
        __syncthreads();
      }
    }
  }

  // ---- cross-wave K reduction through LDS, row-major epilogue (operand loads batched, unconditional) ----
  float *myred = red + (size_t)wave * BM * LDR;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) myred[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + j * 32 + fr] = acc[i][j][r];
  __syncthreads();

  T *out = static_cast<T *>(a.out);
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int idx = tid + it * 256;
    const int ml = idx / QN, nq = idx - ml * QN;
    const int m = m0 + ml, nb = n0 + nq * 4;
    const bool live = idx < BM * QN && m < a.M && nb < a.n_store;
    EpiOps eo;
    if constexpr (HOIST) eo = eo0;
    else eo = epi_load(it);
    const float *bi = eo.bi, *rv = eo.rv, *sv = eo.sv, *av = eo.av;
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

template <typename T, int BM, int BN, bool CAT, int KW, int NSET> hipError_t launch_fast3(const ConvGemmArgs &a, hipStream_t s) {
  constexpr int BKT = 4 * KW;
  constexpr int LD = BKT + 16 / (int)sizeof(T);
  constexpr size_t stage_bytes = (size_t)(BM + BN) * LD * sizeof(T);
  constexpr size_t red_bytes = (size_t)4 * BM * (BN + 4) * sizeof(float);
  const size_t lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.n_store + BN - 1) / BN;
  const int swz = (ntiles % 8 == 0) ? 1 : 0;
  const size_t es = sizeof(T);
  const size_t bA = (size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * es;
  const size_t bA2 = CAT ? (size_t)a.M * a.src2_ld * es : 0;
  const size_t bW = (size_t)a.N * a.K * es;
  auto kern = conv_gemm_fast_kernel<T, BM, BN, CAT, KW, NSET>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(256), lds, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW);
  return hipGetLastError();
}

// Two chunks in flight: measured with HBM-cold weights (tools/gemm_cold.py), 4 or 6 chunks in flight are no faster
// (the limit is the L2 -> CU fill rate, ~25 B/clk/CU here, not latency) and 6 cost occupancy.
template <typename T, int BM, int BN, bool CAT, int KW> hipError_t launch_fast2(const ConvGemmArgs &a, hipStream_t s) {
  return launch_fast3<T, BM, BN, CAT, KW, 2>(a, s);
}

}  // namespace

// eligibility: 1-D, <= 3 taps, no prologue, channel count a multiple of the K chunk, every buffer < 2 GiB
bool conv_gemm_fast_ok(int dt, const ConvGemmArgs &a) {
  if (a.geom != 0 || a.pro != 0 || a.taps > 3 || a.taps < 1) return false;
  if ((a.cin % 128) || (a.cin2 % 32) || (a.K % 32)) return false;
  const size_t es = dsize(dt);
  const size_t lim = 0x7FFFFFF0ull;
  const size_t clips = (size_t)(a.M / a.Lout + 1);
  if (clips * a.Lsrc * a.src_ld * es >= lim) return false;
  if ((size_t)a.M * (a.src2_ld > 0 ? a.src2_ld : 1) * es >= lim) return false;
  if ((size_t)a.N * a.K * es >= lim) return false;
  return true;
}

hipError_t launch_conv_gemm_fast(int dt, const ConvGemmArgs &a, int variant, hipStream_t s) {
  // 256-wide chunks (64 K per wave and iteration: half the barriers) when the channel count allows it
  const bool wide = (a.cin % 256) == 0 && g_conv_gemm_force.sk == 64;   // measured: no gain over 128-wide chunks
#define SF_FAST(T, BM, BN)                                                                                              \
  (wide ? (a.cin2 ? launch_fast2<T, BM, BN, true, 64>(a, s) : launch_fast2<T, BM, BN, false, 64>(a, s))                 \
        : (a.cin2 ? launch_fast2<T, BM, BN, true, 32>(a, s) : launch_fast2<T, BM, BN, false, 32>(a, s)))
  if (dt == F32) {
    switch (variant) {
      case 0: return SF_FAST(float, 64, 64);
      case 1: return SF_FAST(float, 64, 32);
      default: return SF_FAST(float, 32, 32);
    }
  }
  switch (variant) {
    case 0: return SF_FAST(bf16, 64, 64);
    case 1: return SF_FAST(bf16, 64, 32);
    default: return SF_FAST(bf16, 32, 32);
  }
#undef SF_FAST
}

}  // namespace sf
