// Wave-private split-K implicit GEMM ("wp") for SHORT activations: the deep U-Net levels at small batch
// (M = B*L_d = 352..5632 rows, N <= 1536, K = 384..3072).
//
// Measured on the staged wave-split-K kernel (conv_gemm_fast.hip, rocprofv3 PMC, M=352 N=1024 K=3072): each wave
// lives 32 k cycles for 24 K-chunks -- 1330 cycles per chunk for ~75 instructions and 2 MFMAs; 49 % of the wave
// cycles are spent parked at s_waitcnt / s_barrier.  The four waves of a workgroup multiply DIFFERENT K slices, so
// they share no operand: the two workgroup barriers per chunk only coupled four independent streams.
//
// Here every wave runs its own pipeline over its own quarter of K, with NO barrier in the loop:
//   * full-line loads: one buffer_load_dwordx4 covers 8 rows x 128 B (a 64-wide K chunk of bf16 is exactly a line),
//     offsets hoisted (streaming state as in conv_gemm_v2), out-of-range rows / K tails return zero;
//   * two register sets keep the loads of chunks t+1 and t+2 in flight under the MFMAs of chunk t;
//   * the chunk is transposed into MFMA fragment order through a WAVE-PRIVATE LDS region (same-wave LDS operations
//     execute in order: only lgkmcnt waits, no s_barrier);
//   * the four partial tiles meet once, in the epilogue (fixed-order sum through LDS -> deterministic).
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int BK = 64;
constexpr unsigned OOB = 0x80000000u;

template <typename T> __device__ __forceinline__ Vec16<T> buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  Vec16<T> v;
  u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  v.v = __builtin_bit_cast(decltype(v.v), raw);
  return v;
}

// NSET = K chunks a wave keeps in flight (register sets); ONE private LDS staging buffer per wave suffices because a
// wave's LDS operations execute in program order (the next chunk's writes queue behind this chunk's fragment reads).
// X3 (T = float): split-fp16 weights (ConvGemmArgs::wx: per 32 k, 32 hi | 32 lo' -- the same 256 bytes per row and chunk), fp32 activation
// fragments split in registers, three v_mfma_f32_32x32x16_f16 per product (common.h, x3_split)
template <typename T, int BM, int BN, bool CAT, int NSET, int X3 = 0>
__global__ __launch_bounds__(256) void conv_gemm_wp_kernel(const ConvGemmArgs a, const int mtiles, const int ntiles, const int swz,
                                                           const unsigned bytesA, const unsigned bytesA2, const unsigned bytesW) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int ES = (int)sizeof(T);
  constexpr int VPR = BK / VEC;          // vectors per 64-wide row: bf16 8, fp32 16
  constexpr int RPI = 64 / VPR;          // rows one wave-instruction covers: 8 / 4
  constexpr int PA = BM / RPI, PB = BN / RPI;
  constexpr int LD = BK + 16 / ES;       // 144 B / 272 B rows: conflict-free b128 fragment reads
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int LDR = BN + 4;
  constexpr int WSTAGE = (BM + BN) * LD; // elements of one wave-private staging buffer

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T *wlds = reinterpret_cast<T *>(smem) + (size_t)wave * WSTAGE;   // [A rows | W rows]
  float *red = reinterpret_cast<float *>(smem);

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
  const int lrow = lane / VPR, lvec = lane % VPR;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(CAT ? a.src2 : a.src), 0, CAT ? bytesA2 : 0, 0x00020000);
  static_assert(!X3 || sizeof(T) == 4, "split mode: fp32 activations");
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(X3 ? a.wx : a.w), 0, bytesW, 0x00020000);

  // ---- per staged row (once): clip base, first tap position, validity ------------------------------------------
  int rbase[PA], rp0[PA];
  unsigned roff2[PA], woff[PB], vmask[PA];
  const int pmax = (a.Lsrc << a.up_shift) - 1;
  const unsigned lane_b = (unsigned)(lvec * VEC * ES);
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + i * RPI + lrow;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    vmask[i] = vm ? 0u : OOB;
    const int b = mm / a.Lout, l = mm - b * a.Lout;
    rbase[i] = b * a.Lsrc;
    rp0[i] = l * a.stride - a.pad;
    roff2[i] = CAT ? (((unsigned)(mm * a.src2_ld * ES) + lane_b) | vmask[i]) : OOB;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = n0 + i * RPI + lrow;
    woff[i] = n < a.N ? ((unsigned)(n * a.K * ES) + lane_b) : OOB;
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

  // ---- this wave's K range: a contiguous quarter of the 64-wide chunks ----------------------------------------
  const int k_taps = a.taps * a.cin;
  const int nk = (a.K + BK - 1) / BK;
  const int per = (nk + 3) / 4;
  const int c0 = wave * per;
  const int nkt = max(0, min(per, nk - c0));

  unsigned cur[PA], cb, kb;
  int tap;
  bool second = false;
  const unsigned tap_bytes = (unsigned)(a.cin * ES), kbytes = (unsigned)(a.K * ES);
  auto retap = [&](int t) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int p = rp0[i] + t;
      const unsigned bad = ((unsigned)p > (unsigned)pmax) ? OOB : 0u;
      cur[i] = ((unsigned)(((rbase[i] + (max(p, 0) >> a.up_shift)) * a.src_ld) * ES) + lane_b) | bad | vmask[i];
    }
  };
  {
    const int k0 = c0 * BK;
    kb = (unsigned)(k0 * ES);
    if (CAT && k0 >= k_taps) {
      second = true;
      tap = a.taps;
      cb = (unsigned)((k0 - k_taps) * ES);
#pragma unroll
      for (int i = 0; i < PA; ++i) cur[i] = roff2[i];
    } else {
      tap = min(k0 / a.cin, a.taps - 1);
      cb = (unsigned)((k0 - tap * a.cin) * ES);
      retap(tap);
    }
  }

  struct RegSet {
    Vec16<T> ra[PA], rb[PB];
  };
  RegSet rs[NSET];
  auto prefetch = [&](RegSet &R) {
    const unsigned tmask = (kb + lane_b >= kbytes) ? OOB : 0u;
#pragma unroll
    for (int i = 0; i < PB; ++i) R.rb[i] = buf_ld16<T>(rW, (woff[i] + kb) | tmask);
    const __amdgpu_buffer_rsrc_t rs = (CAT && second) ? rA2 : rA;
#pragma unroll
    for (int i = 0; i < PA; ++i) R.ra[i] = buf_ld16<T>(rs, (cur[i] + cb) | tmask);
    kb += (unsigned)(BK * ES);
    cb += (unsigned)(BK * ES);
    if (!second && cb >= tap_bytes) {
      cb = 0;
      ++tap;
      if (tap < a.taps) retap(tap);
      else {
        second = true;
#pragma unroll
        for (int i = 0; i < PA; ++i) cur[i] = roff2[i];
      }
    }
  };
  auto stage = [&](RegSet &R) {
    T *As = wlds, *Bs = As + BM * LD;

#pragma unroll
    for (int i = 0; i < PB; ++i) st16<T>(Bs + (i * RPI + lrow) * LD + lvec * VEC, R.rb[i]);
#pragma unroll
    for (int i = 0; i < PA; ++i) st16<T>(As + (i * RPI + lrow) * LD + lvec * VEC, R.ra[i]);
  };
  const int fr = lane & 31, fh = lane >> 5;
  auto compute = [&]() {
    const T *As = wlds, *Bs = As + BM * LD;
    if constexpr (X3) {
      // the 64-deep chunk = four 16-deep products; weight row: two groups of (32 hi | 32 lo'), product s reads eight hi at byte
      // 128 (s / 2) + 32 (s % 2) + 16 fh and the matching lo' 64 bytes behind
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        using xv = typename X3P<X3 ? X3 : 1>::v8;
        xv ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float *ap = reinterpret_cast<const float *>(As) + (i * 32 + fr) * LD + 16 * s + 8 * fh;
          x3_split<X3 ? X3 : 1>(*reinterpret_cast<const f32x4 *>(ap), *reinterpret_cast<const f32x4 *>(ap + 4), ah[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned char *bp = reinterpret_cast<const unsigned char *>(Bs + (j * 32 + fr) * LD) + 128 * (s >> 1) + 32 * (s & 1) + 16 * fh;
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
      for (int s = 0; s < BK / 16; ++s) {
        using frag = typename Frag16<T>::type;
        frag af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const frag *>(As + (i * 32 + fr) * LD + 16 * s + 8 * fh);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const frag *>(Bs + (j * 32 + fr) * LD + 16 * s + 8 * fh);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32x16(af[i], bfr[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f32x4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4 *>(As + (i * 32 + fr) * LD + 32 * fh + 4 * q);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const f32x4 *>(Bs + (j * 32 + fr) * LD + 32 * fh + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- epilogue operands: loaded up front when one epilogue pass per thread suffices (see conv_gemm_fast.hip) ------
  const T *res = static_cast<const T *>(a.res);
  const bool has_res = res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
  constexpr int QN = BN / 4;
  constexpr int EIT = (BM * QN + 255) / 256;
  constexpr bool HOIST = EIT == 1;
  // LayerNorm of the raw source folded into the accumulator (the qkv projection behind the attention pre-norm, see
  // conv_gemm_fast.hip): out = rstd_m * (acc - mean_m * colsum(W)_n); the row statistics are pooled from the producer's per-tile
  // partials, whose loads ride with the epilogue operands in front of the K loop (32-row tiles only)
  const bool ln_epi = BM == 32 && a.ln_colsum != nullptr;
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
  float ln_mp[4] = {0.f, 0.f, 0.f, 0.f}, ln_qp[4] = {0.f, 0.f, 0.f, 0.f};
  if (ln_epi) {   // 8 threads per row: partial t8 + 8 j of row tid >> 3
    const int m = min(m0 + (tid >> 3), a.M - 1), t8 = tid & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pidx = t8 + 8 * j;
      if (pidx < a.ln_nt) {
        const float2 pv = *reinterpret_cast<const float2 *>(a.ln_part + ((size_t)m * a.ln_nt + pidx) * 2);
        ln_mp[j] = pv.x;
        ln_qp[j] = pv.y;
      }
    }
  }

  // ---- the wave's own pipeline: no workgroup barrier --------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NSET; ++j)
    if (j < nkt) prefetch(rs[j]);
  // Same-wave LDS operations execute in program order, so "write the chunk, then read the fragments" needs no
  // s_barrier; the wave_barrier()s only stop the COMPILER from moving LDS accesses of different lanes across the
  // stage/compute boundaries.
  for (int kt = 0; kt < nkt; kt += NSET) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) {
      if (kt + j < nkt) {
        stage(rs[j]);
        __builtin_amdgcn_wave_barrier();
        if (kt + j + NSET < nkt) prefetch(rs[j]);
        compute();
        __builtin_amdgcn_wave_barrier();
      }
    }
  }

  // ---- the four partial tiles meet here --------------------------------------------------------------------
  __syncthreads();   // every wave is done with its staging buffers (red aliases them)
  float *rowstat = red + (size_t)4 * BM * LDR;   // (mean, rstd) per row, behind the four partial tiles
  float *gsum = rowstat + 2 * BM;                // (sum, sum of squares) per row of the stored tile (gnpart_out)
  if (ln_epi) {
    const int t8 = tid & 7;
    const float mean = sum8_dpp((ln_mp[0] + ln_mp[1]) + (ln_mp[2] + ln_mp[3])) / (float)a.ln_nt;   // every partial covers 32 channels
    float dq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (t8 + 8 * j < a.ln_nt) {
        const float d = ln_mp[j] - mean;
        dq += fmaf(32.f * d, d, ln_qp[j]);
      }
    const float m2 = sum8_dpp(dq);
    if (t8 == 0) {
      rowstat[2 * (tid >> 3)] = mean;
      rowstat[2 * (tid >> 3) + 1] = rsqrtf(m2 / (float)a.cin + a.ln_eps);
    }
  }
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
    if (ln_epi) {
      const float mu = rowstat[2 * mlc], rstd = rowstat[2 * mlc + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rstd * (v[e] - mu * eo.cu[e]);
    }
    float xo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nb + e;
      float x = (v[e] + bi[e]) * sv[e] + rv[e] + av[e];
      x = n < a.N ? apply_act(x, a.act) : 0.f;
      xo[e] = a.out_f32 ? x : to_f(from_f<T>(x));
      if (live && n < a.n_store) {
        if (a.out_f32) static_cast<float *>(a.out)[(size_t)m * a.out_ld + n] = x;
        else out[(size_t)m * a.out_ld + n] = from_f<T>(x);
      }
    }
    if constexpr (HOIST && BN == 32) {
      if (a.rowpart_out) {   // row-LayerNorm partial of the stored values (see ConvGemmArgs, conv_gemm_fast.hip)
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
      if (a.gnpart_out) {   // GroupNorm tile sums of the stored values for the channel-block convolution that follows (kernels.h)
        const float s1 = sum8_dpp((xo[0] + xo[1]) + (xo[2] + xo[3]));
        const float s2 = sum8_dpp(fmaf(xo[0], xo[0], xo[1] * xo[1]) + fmaf(xo[2], xo[2], xo[3] * xo[3]));
        if (nq == 0) {
          gsum[2 * ml] = live ? s1 : 0.f;
          gsum[2 * ml + 1] = live ? s2 : 0.f;
        }
        __syncthreads();
        if (tid < 2) {   // segment 0: rows of the first row's clip; segment 1: rows of the next clip.  Fixed order: deterministic
          const int rb = min((m0 / a.Lout + 1) * a.Lout - m0, BM);
          const int lo = tid == 0 ? 0 : rb, hi = tid == 0 ? rb : BM;
          float t1 = 0.f, t2 = 0.f;
          for (int r = lo; r < hi; ++r) {
            t1 += gsum[2 * r];
            t2 += gsum[2 * r + 1];
          }
          *reinterpret_cast<float2 *>(a.gnpart_out + (((size_t)mt * ntiles + nt) * 2 + tid) * 2) = make_float2(t1, t2);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, bool CAT, int NSET, int X3 = 0> hipError_t launch_wp3(const ConvGemmArgs &a, hipStream_t s) {
  constexpr int LD = BK + 16 / (int)sizeof(T);
  constexpr size_t stage_bytes = (size_t)4 * (BM + BN) * LD * sizeof(T);
  constexpr size_t red_bytes = (size_t)4 * BM * (BN + 4) * sizeof(float) + (size_t)4 * BM * sizeof(float);   // + (mean, rstd) and (sum, sumsq) per row
  const size_t lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  if (lds > 150 * 1024) return hipErrorInvalidValue;   // fp32 64-row tiles do not fit: the caller falls back
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.n_store + BN - 1) / BN;
  const int swz = (ntiles % 8 == 0) ? 1 : 0;
  const size_t es = sizeof(T);
  const size_t bA = (size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * es;
  const size_t bA2 = CAT ? (size_t)a.M * a.src2_ld * es : 0;
  const size_t bW = (size_t)a.N * a.K * es;
  auto kern = conv_gemm_wp_kernel<T, BM, BN, CAT, NSET, X3>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles + (a.pf.ptr && a.pf.bytes >= 16 ? a.pf.wgs : 0)), dim3(256), lds, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW);
  return hipGetLastError();
}

// two chunks in flight per wave: three or four measured slower (register pressure against 2.7 waves per SIMD)
template <typename T, int BM, int BN, bool CAT> hipError_t launch_wp2(const ConvGemmArgs &a, hipStream_t s) {
  return launch_wp3<T, BM, BN, CAT, 2>(a, s);
}

}  // namespace

bool conv_gemm_wp_ok(int dt, const ConvGemmArgs &a) {
  if (a.geom != 0 || a.pro != 0 || a.taps < 1) return false;
  if ((a.cin % BK) || (a.cin2 % 32) || (a.K % 32)) return false;
  const size_t es = dsize(dt), lim = 0x7FFFFFF0ull;
  if ((size_t)(a.M / a.Lout + 1) * a.Lsrc * a.src_ld * es >= lim) return false;
  if ((size_t)a.M * (a.src2_ld > 0 ? a.src2_ld : 1) * es >= lim) return false;
  if ((size_t)a.N * a.K * es >= lim) return false;
  return true;
}

// variant: 0 = 64x64, 1 = 64x32, 2 = 32x32
hipError_t launch_conv_gemm_wp(int dt, const ConvGemmArgs &a, int variant, hipStream_t s) {
#define SF_WP(T, BM, BN) (a.cin2 ? launch_wp2<T, BM, BN, true>(a, s) : launch_wp2<T, BM, BN, false>(a, s))
  if (dt == F32 && a.wx) {   // split mode: 32x32 tiles (the variants the fp32 engine uses on short activations)
    if (a.wx_mode == X3_BF16) return a.cin2 ? hipErrorInvalidValue : launch_wp3<float, 32, 32, false, 2, X3_BF16>(a, s);
    return a.cin2 ? launch_wp3<float, 32, 32, true, 2, X3_F16>(a, s) : launch_wp3<float, 32, 32, false, 2, X3_F16>(a, s);
  }
  if (dt == F32) {
    switch (variant) {
      case 0: return SF_WP(float, 64, 64);
      case 1: return SF_WP(float, 64, 32);
      default: return SF_WP(float, 32, 32);
    }
  }
  if (dt == F16) {
    switch (variant) {
      case 0: return SF_WP(f16, 64, 64);
      case 1: return SF_WP(f16, 64, 32);
      default: return SF_WP(f16, 32, 32);
    }
  }
  switch (variant) {
    case 0: return SF_WP(bf16, 64, 64);
    case 1: return SF_WP(bf16, 64, 32);
    default: return SF_WP(bf16, 32, 32);
  }
#undef SF_WP
}

}  // namespace sf
