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
#include <cstdlib>

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
// LN: the first source is LayerNorm-modulated on the fly (ConvGemmArgs::ln_*; 32x32 tiles, one tap, Lout >= 32).
// X3 (T = float, KW = 32): split-fp16 weights (ConvGemmArgs::wx; wave w multiplies the w-th (32 hi | 32 lo') group of the staged 128-deep
// chunk), fp32 activation fragments split in registers after the (optional) LayerNorm transform, three fp16 MFMAs per product (common.h)
template <typename T, int BM, int BN, bool CAT, int KW, int NSET, bool LN, int X3 = 0>
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
  // LN: behind the staging / reduction area: rowstat[BM][2] = (mean, rstd), tab[2 clips][2][cin] = (1 + scale | shift)
  constexpr size_t kStageBytes = (size_t)(BM + BN) * LD * sizeof(T), kRedBytes = (size_t)4 * BM * LDR * sizeof(float);
  constexpr size_t kLnOff = ((kStageBytes > kRedBytes ? kStageBytes : kRedBytes) + 15) / 16 * 16;
  float *rowstat = reinterpret_cast<float *>(smem + kLnOff);
  float *lntab = rowstat + 2 * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= mtiles * ntiles) {   // hosted weight prefetch for the next GEMM of the chain (kernels.h, Prefetch)
    prefetch_slice(a.pf, (int)blockIdx.x - mtiles * ntiles, 256);
    return;
  }
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
  static_assert(!X3 || (sizeof(T) == 4 && KW == 32), "split mode: fp32 activations, one 32-deep group per wave");
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(X3 ? a.wx : a.w), 0, bytesW, 0x00020000);

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
  f32x16 accL[X3 ? TM : 1][X3 ? TN : 1];   // split mode: cross terms (scaled by 2048)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (X3) accL[i][j][r] = 0.f;
      }

  // LN: per staged row (mean, rstd, clip selector) -- filled after the first chunks are in flight
  float ln_mu[PA], ln_rs[PA];
  int ln_cs[PA];
  const int ln_b0 = m0 / a.Lout;
  const bool has_ss = LN && a.ln_ss != nullptr;
  const bool ln_epi = LN && a.ln_colsum != nullptr;   // normalise the accumulator instead of the operand
  struct RegSet {
    Vec16<T> ra[PA], rb[PB];
    unsigned c0;   // LN: first channel of the chunk inside the first source
    bool first;    // LN: the chunk belongs to the first source
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
    if constexpr (LN) {
      R.c0 = cb / (unsigned)ES;
      R.first = !second;
    }
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
    if constexpr (LN) {
      if (R.first && !ln_epi) {   // wave-uniform: y = x * (rstd * sc) + (sh - mean * rstd * sc), table rows read as 16-byte vectors
        const int c = (int)R.c0 + svec * VEC;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          Vec16<T> v = R.ra[i];
          if (has_ss) {
            const f32x4 *tsc = reinterpret_cast<const f32x4 *>(lntab + (ln_cs[i] * 2 + 0) * a.cin + c);
            const f32x4 *tsh = reinterpret_cast<const f32x4 *>(lntab + (ln_cs[i] * 2 + 1) * a.cin + c);
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
              const f32x4 sc4 = tsc[q], sh4 = tsh[q];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float A = ln_rs[i] * sc4[e];
                v.set(4 * q + e, fmaf(v.get(4 * q + e), A, fmaf(-ln_mu[i], A, sh4[e])));
              }
            }
          } else {
            const float B = -ln_mu[i] * ln_rs[i];
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, fmaf(v.get(j), ln_rs[i], B));
          }
          st16<T>(As + (i * RPP + srow) * LD + svec * VEC, v);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) st16<T>(As + (i * RPP + srow) * LD + svec * VEC, R.ra[i]);
  };

  const int fr = lane & 31, fh = lane >> 5;
  const int kw0 = KW * wave;
  auto compute = [&]() {
    if constexpr (X3) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        using xv = typename X3P<X3 ? X3 : 1>::v8;
        xv ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float *ap = reinterpret_cast<const float *>(As) + (i * 32 + fr) * LD + kw0 + 16 * s + 8 * fh;
          x3_split<X3 ? X3 : 1>(*reinterpret_cast<const f32x4 *>(ap), *reinterpret_cast<const f32x4 *>(ap + 4), ah[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned char *bp = reinterpret_cast<const unsigned char *>(Bs + (j * 32 + fr) * LD + kw0) + 32 * s + 16 * fh;
          bh[j] = *reinterpret_cast<const xv *>(bp);
          bl[j] = *reinterpret_cast<const xv *>(bp + 64);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) x3_mfma<X3 ? X3 : 1>(ah[i], al[i], bh[j], bl[j], acc[i][j], accL[i][j]);
      }
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < KW / 16; ++s) {
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
  constexpr bool HOIST = EIT == 1 && !LN;   // (the LN variant would pass 128 registers and lose a workgroup per CU)
  struct EpiOps {
    float bi[4], rv[4], sv[4], av[4], cu[4];
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
      o.cu[e] = ln_epi ? a.ln_colsum[nc] : 0.f;
    }
    return o;
  };
  EpiOps eo0;
  if constexpr (HOIST) eo0 = epi_load(0);

#pragma unroll
  for (int j = 0; j < NSET; ++j)
    if (j < nkt) prefetch(rs[j]);
  // ---- LN: pooled row statistics from the producer's per-tile partials, modulation table -> LDS --------------------
  if constexpr (LN) {
    {
      const int r = tid >> 3, t8 = tid & 7;           // 8 threads per row, 32 rows
      const int m = min(m0 + r, a.M - 1);
      float mp[4], qp[4];
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pidx = t8 + 8 * j;
        const bool on = pidx < a.ln_nt;
        const float2 pv = on ? *reinterpret_cast<const float2 *>(a.ln_part + ((size_t)m * a.ln_nt + pidx) * 2) : make_float2(0.f, 0.f);
        mp[j] = pv.x;
        qp[j] = pv.y;
        sm += pv.x;
      }
      const float mean = sum8_dpp(sm) / (float)a.ln_nt;   // every partial covers 32 channels
      float dq = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (t8 + 8 * j < a.ln_nt) {
          const float d = mp[j] - mean;
          dq += fmaf(32.f * d, d, qp[j]);
        }
      }
      const float m2 = sum8_dpp(dq);
      if (t8 == 0) {
        rowstat[2 * r] = mean;
        rowstat[2 * r + 1] = rsqrtf(m2 / (float)a.cin + a.ln_eps);
      }
    }
    if (a.ln_ss) {
      const int nclip = (a.M + a.Lout - 1) / a.Lout;
      for (int i = tid; i < 2 * a.cin; i += 256) {
        const int cs = i >= a.cin, c = i - cs * a.cin;
        const int b = min(ln_b0 + cs, nclip - 1);
        lntab[(cs * 2 + 0) * a.cin + c] = 1.0f + a.ln_ss[(size_t)b * a.ln_ss_ld + c];
        lntab[(cs * 2 + 1) * a.cin + c] = a.ln_ss[(size_t)b * a.ln_ss_ld + a.cin + c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int ml = i * RPP + srow;
      ln_mu[i] = rowstat[2 * ml];
      ln_rs[i] = rowstat[2 * ml + 1];
      ln_cs[i] = (min(m0 + ml, a.M - 1) / a.Lout != ln_b0) ? 1 : 0;
    }
  }

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
      for (int r = 0; r < 16; ++r) {
        if constexpr (X3) myred[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + j * 32 + fr] = fmaf(accL[i][j][r], X3P<X3 ? X3 : 1>::INV, acc[i][j][r]);
        else myred[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + j * 32 + fr] = acc[i][j][r];
      }
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
    float rvt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rvt[e] = rv[e];
    if constexpr (LN) {
      if (ln_epi) {   // LayerNorm of the raw source folded into the accumulator
        const float mu = rowstat[2 * mlc], rstd = rowstat[2 * mlc + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rstd * (v[e] - mu * eo.cu[e]);
      }
    }
    if constexpr (LN) {
      if (a.res_ln) {   // the residual is the LayerNorm-modulated source row itself
        const float mu = rowstat[2 * mlc], rstd = rowstat[2 * mlc + 1];
        const int cs = (min(m, a.M - 1) / a.Lout != ln_b0) ? 1 : 0;
        const int nbc = min(nb, a.N - 4);   // res_ln implies N == cin, a multiple of 32
        const f32x4 sc4 = has_ss ? *reinterpret_cast<const f32x4 *>(lntab + (cs * 2 + 0) * a.cin + nbc) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh4 = has_ss ? *reinterpret_cast<const f32x4 *>(lntab + (cs * 2 + 1) * a.cin + nbc) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) rvt[e] = fmaf((rv[e] - mu) * rstd, sc4[e], sh4[e]);
      }
    }
    float xo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nb + e;
      float x = (v[e] + bi[e]) * sv[e] + rvt[e] + av[e];
      x = n < a.N ? apply_act(x, a.act) : 0.f;
      xo[e] = a.out_f32 ? x : to_f(from_f<T>(x));
      if (live && n < a.n_store) {
        if (a.out_f32) static_cast<float *>(a.out)[(size_t)m * a.out_ld + n] = x;
        else out[(size_t)m * a.out_ld + n] = from_f<T>(x);
      }
    }
    if constexpr (EIT == 1 && BN == 32) {
      if (a.rowpart_out) {   // (mean, M2) of this row's 32 stored values: 8 consecutive lanes hold them, 4 each
        const float mean = sum8_dpp((xo[0] + xo[1]) + (xo[2] + xo[3])) * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xo[e] - mean;
          q = fmaf(d, d, q);
        }
        q = sum8_dpp(q);
        if (nq == 0 && live) *reinterpret_cast<float2 *>(a.rowpart_out + ((size_t)m * a.rowpart_nt + nt) * 2) = make_float2(mean, q);
      }
    }
  }
}

template <typename T, int BM, int BN, bool CAT, int KW, int NSET, bool LN = false, int X3 = 0> hipError_t launch_fast3(const ConvGemmArgs &a, hipStream_t s) {
  constexpr int BKT = 4 * KW;
  constexpr int LD = BKT + 16 / (int)sizeof(T);
  constexpr size_t stage_bytes = (size_t)(BM + BN) * LD * sizeof(T);
  constexpr size_t red_bytes = (size_t)4 * BM * (BN + 4) * sizeof(float);
  size_t lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  if (LN) lds = (lds + 15) / 16 * 16 + (size_t)(2 * BM + (a.ln_ss ? 4 * a.cin : 0)) * sizeof(float);   // rowstat (+ modulation table)
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.n_store + BN - 1) / BN;
  const int swz = (ntiles % 8 == 0) ? 1 : 0;
  const size_t es = sizeof(T);
  const size_t bA = (size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * es;
  const size_t bA2 = CAT ? (size_t)a.M * a.src2_ld * es : 0;
  const size_t bW = (size_t)a.N * a.K * es;
  auto kern = conv_gemm_fast_kernel<T, BM, BN, CAT, KW, NSET, LN, X3>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles + (a.pf.ptr && a.pf.bytes >= 16 ? a.pf.wgs : 0)), dim3(256), lds, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW);
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

// long activations: the macro-tile GEMM beats the LayerNorm-fused 32x32 kernel; it carries the accumulator-side LayerNorm (ln_colsum)
// but no operand transform (Modulation + InjectChannels keep their ln_modulate launch there)
static bool ln_goes_mt(int dt, const ConvGemmArgs &a, bool *supported = nullptr) {
  ConvGemmArgs plain = a;
  plain.ln_part = nullptr;
  plain.ln_colsum = nullptr;
  plain.ln_ss = nullptr;
  plain.rowpart_out = nullptr;
  plain.res_ln = 0;
  if (!conv_gemm_mt_wanted(dt, plain)) return false;
  const bool off = a.mt_ln == 0;
  if (supported) *supported = !off && a.ln_part && a.ln_colsum && !a.ln_ss && !a.res_ln && conv_gemm_mt_ok(dt, a);
  return true;
}

bool conv_gemm_ln_ok(int dt, const ConvGemmArgs &a) {
  {
    bool sup = false;
    if (ln_goes_mt(dt, a, &sup)) return sup;
  }
  if (!conv_gemm_fast_ok(dt, a)) return false;
  // the operand-transform form runs on the 32x32 staged kernel: short activations only (with the macro-tile row partials a long
  // producer can offer them too -- depth 3 at batch 32: 8 launches of 13 us more than ln_modulate + the 64x64 kernel)
  if ((a.ln_ss || a.res_ln || !a.ln_colsum) && (long)((a.M + 63) / 64) * ((a.n_store + 63) / 64) >= 500) return false;
  if (a.taps != 1 || a.stride != 1 || a.up_shift != 0 || a.Lout != a.Lsrc || a.Lout < 32) return false;
  if (!a.ln_part || a.ln_nt * 32 != a.cin || a.ln_nt > 32) return false;
  if (a.res_ln && (a.N != a.cin || !a.res)) return false;
  if (a.ln_colsum && (a.cin2 || a.ln_ss || a.res_ln)) return false;
  if (a.rowpart_out && ((a.n_store % 32) || a.rowpart_nt * 32 != a.n_store)) return false;
  return true;
}

bool conv_gemm_wp_ok(int dt, const ConvGemmArgs &a);                                           // conv_gemm_wp.hip
hipError_t launch_conv_gemm_wp(int dt, const ConvGemmArgs &a, int variant, hipStream_t s);

static bool ln_goes_wp(int dt, const ConvGemmArgs &a) {
  // tuning hook: the wave-private kernel also carries the epilogue fold, but on the qkv projections of this model (288 tiles,
  // K = 1024) the staged kernel measured 1.2 % faster over a whole step (460 vs 455 steps/s), so it stays opt-in
  static const bool use_wp = tune_env("SF_LN_WP") != nullptr;
  return use_wp && a.ln_colsum && !a.ln_ss && !a.res_ln && !a.rowpart_out && conv_gemm_wp_ok(dt, a);
}

static bool ln_goes_rs(int dt, const ConvGemmArgs &a);
const char *conv_gemm_ln_variant_name(int dt, const ConvGemmArgs &a) {
  if (ln_goes_mt(dt, a)) return label_for_dtype(dt, conv_gemm_mt_name(a));
  if (ln_goes_rs(dt, a)) return dt == F32 ? "conv_gemm_rs<x3,32x32>" : label_for_dtype(dt, "conv_gemm_rs<bf16,32x32>");
  if (dt == F32 && a.wx && a.wx_mode == X3_F16) return ln_goes_wp(dt, a) ? "conv_gemm_wp<x3,32x32>" : "conv_gemm_fast<x3,32x32>";
  if (dt == F32) return ln_goes_wp(dt, a) ? "conv_gemm_wp<f32,32x32>" : "conv_gemm_fast<f32,32x32>";
  return label_for_dtype(dt, ln_goes_wp(dt, a) ? "conv_gemm_wp<bf16,32x32>" : "conv_gemm_fast<bf16,32x32>");
}

// the accumulator-side LayerNorm (raw rows through the matrix cores, rstd * (acc - mean * colsum) in the epilogue) on the register-staged
// kernel when the fragment-ordered weights are at hand and the launch has few tiles
bool conv_gemm_prefers_wp(const ConvGemmArgs &a);
static bool ln_goes_rs(int dt, const ConvGemmArgs &a) {
  static const long max_tiles = [] {   // tuning hook: most 32x32 tiles a LayerNorm-folded projection may have and still take the register-staged kernel
    const char *e = tune_env("SF_RS_LN_TILES");
    return e ? atol(e) : 512L;
  }();
  const long tiles = (long)((a.M + 31) / 32) * ((a.n_store + 31) / 32);
  return a.ln_colsum && !a.ln_ss && !a.res_ln && g_conv_gemm_force.path == 0 && tiles <= max_tiles && a.K >= 256 && conv_gemm_rs_ok(dt, a);
}

hipError_t launch_conv_gemm_ln(int dt, const ConvGemmArgs &a, hipStream_t s) {
  if (!conv_gemm_ln_ok(dt, a)) return hipErrorInvalidValue;
  if (ln_goes_mt(dt, a)) return launch_conv_gemm_mt(dt, a, s);
  if (ln_goes_rs(dt, a)) return launch_conv_gemm_rs(dt, a, s);
  if (ln_goes_wp(dt, a)) return launch_conv_gemm_wp(dt, a, 2, s);
  if (dt == F32 && a.wx && a.wx_mode == X3_F16) return a.cin2 ? launch_fast3<float, 32, 32, true, 32, 2, true, X3_F16>(a, s) : launch_fast3<float, 32, 32, false, 32, 2, true, X3_F16>(a, s);
  return SF_DISPATCH_T(dt, (a.cin2 ? launch_fast3<T, 32, 32, true, 32, 2, true>(a, s) : launch_fast3<T, 32, 32, false, 32, 2, true>(a, s)));
}

hipError_t launch_conv_gemm_fast(int dt, const ConvGemmArgs &a, int variant, hipStream_t s) {
  // 256-wide chunks (64 K per wave and iteration: half the barriers) when the channel count allows it
  const bool wide = (a.cin % 256) == 0 && g_conv_gemm_force.sk == 64;   // measured: no gain over 128-wide chunks
#define SF_FAST(T, BM, BN)                                                                                              \
  (wide ? (a.cin2 ? launch_fast2<T, BM, BN, true, 64>(a, s) : launch_fast2<T, BM, BN, false, 64>(a, s))                 \
        : (a.cin2 ? launch_fast2<T, BM, BN, true, 32>(a, s) : launch_fast2<T, BM, BN, false, 32>(a, s)))
  if (dt == F32 && a.wx && a.wx_mode == X3_BF16 && !a.cin2) {   // split mode, gradients
    switch (variant) {
      case 0: return launch_fast3<float, 64, 64, false, 32, 2, false, X3_BF16>(a, s);
      case 1: return launch_fast3<float, 64, 32, false, 32, 2, false, X3_BF16>(a, s);
      default: return launch_fast3<float, 32, 32, false, 32, 2, false, X3_BF16>(a, s);
    }
  }
  if (dt == F32 && a.wx && a.wx_mode == X3_F16) {   // split mode
#define SF_FASTX(BM, BN) (a.cin2 ? launch_fast3<float, BM, BN, true, 32, 2, false, X3_F16>(a, s) : launch_fast3<float, BM, BN, false, 32, 2, false, X3_F16>(a, s))
    switch (variant) {
      case 0: return SF_FASTX(64, 64);
      case 1: return SF_FASTX(64, 32);
      default: return SF_FASTX(32, 32);
    }
#undef SF_FASTX
  }
  if (dt == F32) {
    switch (variant) {
      case 0: return SF_FAST(float, 64, 64);
      case 1: return SF_FAST(float, 64, 32);
      default: return SF_FAST(float, 32, 32);
    }
  }
  if (dt == F16) {
    switch (variant) {
      case 0: return SF_FAST(f16, 64, 64);
      case 1: return SF_FAST(f16, 64, 32);
      default: return SF_FAST(f16, 32, 32);
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
