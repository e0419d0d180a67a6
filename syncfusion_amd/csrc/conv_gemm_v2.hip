// Main MFMA implicit-GEMM convolution ("v2"): 1-D and video geometry, channel counts that are multiples of 64.
//
//   * 256 threads = 4 waves as 2x2; block tile BM x BN (128x128, 128x64, 64x64), wave tile (BM/2)x(BN/2) of
//     32x32 MFMA tiles; K walked in 64-wide chunks (4 MFMA k-steps per tile per chunk for bf16).
//   * A and W chunks are fetched with `buffer_load_dwordx4` (wave-uniform descriptors, 32-bit per-lane offsets,
//     out-of-range offsets return zero: padding rows, M/N tails and K tails need no predication), issued for
//     chunk t+1 BEFORE the MFMAs of chunk t and written to the OTHER LDS buffer after them: one barrier per chunk.
//   * (a grid split-K variant -- partial tiles in a slab, arrival tickets with agent-scope release / acquire, last
//     arriver reduces -- was measured slower than the wave-split-K kernels on every short-activation shape, 2-7 us of
//     fences per workgroup, and removed.)
//   * epilogue through LDS: row-major 16-byte stores, operand loads (bias / residual / per-clip scale+add)
//     batched and unconditional.
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

struct V2Extra {
  int mtiles, ntiles, swz;
  unsigned bytesA, bytesA2, bytesW;
};

// NSET (even) = K chunks in flight: register sets filled NSET chunks ahead of the MFMAs that consume them.
template <typename T, int BM, int BN, int GEOM, bool CAT, int NSET>
__global__ __launch_bounds__(256) void conv_gemm_v2_kernel(const ConvGemmArgs a, const V2Extra x) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int ES = (int)sizeof(T);
  constexpr int VPR = BK / VEC;        // 16-byte vectors per staged row: bf16 8, fp32 16
  constexpr int RPP = 256 / VPR;       // rows per pass: 32 / 16
  constexpr int PA = BM / RPP, PB = BN / RPP;
  constexpr int LD = BK + 16 / ES;     // padded LDS row (144 B bf16 / 272 B fp32): conflict-free b128 fragment reads
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDR = BN + 4;
  constexpr int STAGE = (BM + BN) * LD;   // elements per LDS buffer

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *lds = reinterpret_cast<T *>(smem);
  float *red = reinterpret_cast<float *>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // ---- block -> (row tile, column tile): all users of one W panel on one XCD ----------------------------
  int mt, nt;
  {
    const int bid = blockIdx.x;
    if (x.swz) {
      const int xcd = bid & 7, j = bid >> 3;
      mt = j % x.mtiles;
      nt = xcd + 8 * (j / x.mtiles);
    } else {
      mt = bid % x.mtiles;
      nt = bid / x.mtiles;
    }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int srow = tid / VPR, svec = tid % VPR;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, x.bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(CAT ? a.src2 : a.src), 0, CAT ? x.bytesA2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.w), 0, x.bytesW, 0x00020000);

  // ---- per staged row: source coordinates (once) ---------------------------------------------------------
  int rbase[PA], rp0[PA], rh[PA], rw_[PA];
  unsigned roff2[PA], woff[PB];
  unsigned vmask[PA];   // 0 for a live output row, OOB for rows beyond M
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + i * RPP + srow;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    vmask[i] = vm ? 0u : OOB;
    if constexpr (GEOM == 0) {
      const int b = mm / a.Lout, l = mm - b * a.Lout;
      rbase[i] = b * a.Lsrc;
      rp0[i] = l * a.stride - a.pad;
      rh[i] = rw_[i] = 0;
    } else {
      const int w_ = mm % a.Wo;
      int r = mm / a.Wo;
      const int h_ = r % a.Ho;
      r /= a.Ho;
      const int t_ = r % a.To, n_ = r / a.To;
      rbase[i] = n_ * a.Ti;
      rp0[i] = t_ * a.st - a.pt;
      rh[i] = h_ * a.sh - a.ph;
      rw_[i] = w_ * a.sw - a.pw;
    }
    roff2[i] = CAT ? ((unsigned)((mm * a.src2_ld + svec * VEC) * ES) | vmask[i]) : OOB;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = n0 + i * RPP + srow;
    woff[i] = n < a.N ? (unsigned)((n * a.K + svec * VEC) * ES) : OOB;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int k_taps = a.taps * a.cin;
  const int nk_total = (a.K + BK - 1) / BK;
  const int kt0 = 0;
  const int nkt = nk_total;
  const unsigned lane_k = (unsigned)(svec * VEC);
  const int pmax = (a.Lsrc << a.up_shift) - 1;

  // two register sets: the loads of chunk t+2 are issued under the MFMAs of chunk t and written to LDS after the
  // MFMAs of chunk t+1 -- two compute phases of latency cover (named sets: static register indexing)
  struct RegSet {
    Vec16<T> ra[PA], rb[PB];
  };
  RegSet rs[NSET];

  // Streaming gather state (chunks are requested in K order).  The expensive per-row arithmetic (clamps, shifts,
  // multiplies, range checks) runs only when the TAP changes -- once every cin/64 chunks; inside a tap a chunk costs
  // one add + one or per staged vector.  Measured before this: 206 VALU instructions per 16 MFMAs (issue-bound).
  unsigned cur[PA];      // byte offset of staged row i at channel 0 of the current tap (OOB bit set when out of range)
  unsigned cb, kb;       // byte offset of the chunk inside the tap's channels / inside a W row
  int tap;
  bool second = false;   // reading the concatenated second source
  const unsigned tap_bytes = (unsigned)(a.cin * ES), kbytes = (unsigned)(a.K * ES), lane_b = lane_k * ES;
  auto retap = [&](int t) {
    if constexpr (GEOM == 0) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int p = rp0[i] + t;
        const unsigned bad = ((unsigned)p > (unsigned)pmax) ? OOB : 0u;
        cur[i] = ((unsigned)(((rbase[i] + (max(p, 0) >> a.up_shift)) * a.src_ld) * ES) + lane_b) | bad | vmask[i];
      }
    } else {
      const int dw = t % a.kw;
      const int r = t / a.kw;
      const int dh = r % a.kh, dt = r / a.kh;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int ti = rp0[i] + dt, hi = rh[i] + dh, wi = rw_[i] + dw;
        const bool ok = (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
        cur[i] = ((unsigned)(((((rbase[i] + max(ti, 0)) * a.Hi + max(hi, 0)) * a.Wi + max(wi, 0)) * a.src_ld) * ES) + lane_b) |
                 (ok ? 0u : OOB) | vmask[i];
      }
    }
  };
  {   // position the stream at the first chunk
    const int k0 = kt0 * BK;
    kb = (unsigned)(k0 * ES);
    if (CAT && k0 >= k_taps) {
      second = true;
      tap = a.taps;
      cb = (unsigned)((k0 - k_taps) * ES);
#pragma unroll
      for (int i = 0; i < PA; ++i) cur[i] = roff2[i];
    } else {
      tap = k0 / a.cin;
      cb = (unsigned)((k0 - tap * a.cin) * ES);
      retap(tap);
    }
  }
  auto prefetch = [&](RegSet &R) {
    const unsigned tmask = (kb + lane_b >= kbytes) ? OOB : 0u;   // K tail of the last chunk
#pragma unroll
    for (int i = 0; i < PB; ++i) R.rb[i] = buf_ld16<T>(rW, (woff[i] + kb) | tmask);
    const __amdgpu_buffer_rsrc_t rs = (CAT && second) ? rA2 : rA;
#pragma unroll
    for (int i = 0; i < PA; ++i) R.ra[i] = buf_ld16<T>(rs, (cur[i] + cb) | tmask);
    kb += (unsigned)(BK * ES);
    cb += (unsigned)(BK * ES);
    if (!second && cb >= tap_bytes) {   // wave-uniform; no memory operation inside
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
  auto stage = [&](int buf, RegSet &R) {
    Vec16<T>(&ra)[PA] = R.ra;
    Vec16<T>(&rb)[PB] = R.rb;
    T *As = lds + buf * STAGE, *Bs = As + BM * LD;
#pragma unroll
    for (int i = 0; i < PB; ++i) st16<T>(Bs + (i * RPP + srow) * LD + svec * VEC, rb[i]);
#pragma unroll
    for (int i = 0; i < PA; ++i) st16<T>(As + (i * RPP + srow) * LD + svec * VEC, ra[i]);
  };

  const int fr = lane & 31, fh = lane >> 5;
  auto compute = [&](int buf) {
    const T *As = lds + buf * STAGE + (wr * WTM) * LD, *Bs = lds + buf * STAGE + (BM + wc * WTN) * LD;
    if constexpr (sizeof(T) == 2) {
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
      // fp32: v_mfma_f32_32x32x2_f32; the two k of a step are {s, 32 + s} of the chunk (same permutation for A and W)
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

  // ---- K loop: loads of chunk t+1 in flight under the MFMAs of chunk t; one barrier per chunk ---------------
  if constexpr (NSET == 1) {
    // one register set, one LDS buffer, two barriers per chunk: fewest registers / least LDS -> most workgroups per CU,
    // which hide each other's load latency
    prefetch(rs[0]);
    stage(0, rs[0]);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) prefetch(rs[0]);
      compute(0);
      __syncthreads();
      if (kt + 1 < nkt) stage(0, rs[0]);
      __syncthreads();
    }
  } else {
#pragma unroll
  for (int j = 0; j < NSET; ++j)
    if (j < nkt) prefetch(rs[j]);
  stage(0, rs[0]);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += NSET) {
#pragma unroll
    for (int j = 0; j < NSET; ++j) {
      const int c = kt + j;   // chunk c sits in LDS buffer j & 1; its register set is free again
      if (c < nkt) {
        if (c + NSET < nkt) prefetch(rs[j]);
        compute(j & 1);
        if (c + 1 < nkt) stage((j + 1) & 1, rs[(j + 1) % NSET]);
        __syncthreads();
      }
    }
  }
  }

  // ---- accumulators -> LDS (fp32, row-major) --------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        red[(wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + wc * WTN + j * 32 + fr] = acc[i][j][r];
  __syncthreads();

  constexpr int QN = BN / 4;
  constexpr int ITER = (BM * QN) / 256;
  T *out = static_cast<T *>(a.out);
  const T *res = static_cast<const T *>(a.res);
  const bool has_res = res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * 256;
    const int ml = idx / QN, nq = idx - ml * QN;
    const int m = m0 + ml, nb = n0 + nq * 4;
    const bool live = m < a.M && nb < a.n_store;
    const int mc = min(m, a.M - 1);
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
    const f32x4 v = *reinterpret_cast<const f32x4 *>(red + (size_t)ml * LDR + nq * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nb + e;
      float xv = (v[e] + bi[e]) * sv[e] + rv[e] + av[e];
      xv = n < a.N ? apply_act(xv, a.act) : 0.f;
      if (live && n < a.n_store) {
        if (a.out_f32) static_cast<float *>(a.out)[(size_t)m * a.out_ld + n] = xv;
        else out[(size_t)m * a.out_ld + n] = from_f<T>(xv);
      }
    }
  }
}

template <typename T, int BM, int BN, int GEOM, bool CAT, int NSET> hipError_t launch_v2_n(const ConvGemmArgs &a, const V2Plan &pl, hipStream_t s) {
  constexpr int LD = BK + 16 / (int)sizeof(T);
  constexpr size_t stage_bytes = (size_t)(NSET == 1 ? 1 : 2) * (BM + BN) * LD * sizeof(T);
  constexpr size_t red_bytes = (size_t)BM * (BN + 4) * sizeof(float) + 16;
  const size_t lds = stage_bytes > red_bytes ? stage_bytes : red_bytes;
  V2Extra x;
  x.mtiles = (a.M + BM - 1) / BM;
  x.ntiles = (a.n_store + BN - 1) / BN;
  x.swz = (x.ntiles % 8 == 0) ? 1 : 0;
  const size_t es = sizeof(T);
  if (a.geom == 0) x.bytesA = (unsigned)((size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * es);
  else x.bytesA = (unsigned)((size_t)(a.M / (a.To * a.Ho * a.Wo)) * a.Ti * a.Hi * a.Wi * a.src_ld * es);
  x.bytesA2 = CAT ? (unsigned)((size_t)a.M * a.src2_ld * es) : 0u;
  x.bytesW = (unsigned)((size_t)a.N * a.K * es);
  auto kern = conv_gemm_v2_kernel<T, BM, BN, GEOM, CAT, NSET>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  hipLaunchKernelGGL(kern, dim3(x.mtiles * x.ntiles), dim3(256), lds, s, a, x);
  return hipGetLastError();
}

// bf16: ONE register set and ONE LDS buffer (two barriers per chunk).  These tiles are bound by the L2 -> CU fill rate
// (~29 B/clk/CU; a 128x128 tile needs 64 B/clk and a 64x64 tile 127 B/clk to keep the matrix cores busy), so what counts is how
// many workgroups a CU holds to overlap each other's loads: 54-92 registers instead of 88-192 give 7 / 4 / 3 workgroups per CU
// for the 64x64 / 128x64 / 128x128 tiles and +15-45 % on every MFMA-bound shape (tools/gemm_big.py); two or four chunks in
// flight per workgroup measured the same or slower.  fp32 (parity path) keeps the double-buffered form.
template <typename T, int BM, int BN, int GEOM, bool CAT> hipError_t launch_v2_t(const ConvGemmArgs &a, const V2Plan &pl, hipStream_t s) {
  if constexpr (sizeof(T) == 2) return launch_v2_n<T, BM, BN, GEOM, CAT, 1>(a, pl, s);
  else return launch_v2_n<T, BM, BN, GEOM, CAT, 2>(a, pl, s);
}

template <typename T, int BM, int BN> hipError_t launch_v2_g(const ConvGemmArgs &a, const V2Plan &pl, hipStream_t s) {
  if (a.geom == 1) return a.cin2 ? hipErrorInvalidValue : launch_v2_t<T, BM, BN, 1, false>(a, pl, s);
  return a.cin2 ? launch_v2_t<T, BM, BN, 0, true>(a, pl, s) : launch_v2_t<T, BM, BN, 0, false>(a, pl, s);
}

}  // namespace

// eligibility + tile choice
bool conv_gemm_v2_plan(int dt, const ConvGemmArgs &a, V2Plan &pl) {
  if (a.pro != 0 || (a.cin % BK) || (a.cin2 % 32) || (a.K % 32) || a.n_store % 4) return false;
  if (a.geom == 1 && a.cin2) return false;
  const size_t es = dsize(dt), lim = 0x7FFFFFF0ull;
  size_t bA;
  if (a.geom == 0) bA = (size_t)(a.M / a.Lout + 1) * a.Lsrc * a.src_ld * es;
  else bA = (size_t)(a.M / (a.To * a.Ho * a.Wo) + 1) * a.Ti * a.Hi * a.Wi * a.src_ld * es;
  if (bA >= lim || (size_t)a.M * (a.src2_ld > 0 ? a.src2_ld : 1) * es >= lim || (size_t)a.N * a.K * es >= lim) return false;
  auto tiles = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.n_store + bn - 1) / bn); };
  // Measured on MI355X (tools/gemm_sweep.py, bf16): 128x128 wins once it yields >= ~300 workgroups, 64x64 otherwise;
  // below ~500 64x64-tiles the wave-split-K kernel with 32x32 tiles is faster (see launch_conv_gemm).
  // 64x64 wins almost everywhere once seven workgroups fit a CU; 128x64 where few row tiles meet wide outputs and a long K
  // (deep U-Net levels at the guidance batch); fp32 keeps the old rule (two-buffer form, parity path only)
  if (dt == F32) pl.variant = (a.n_store >= 128 && tiles(128, 128) >= 300) ? 0 : 2;
  else pl.variant = (a.M <= 8192 && a.n_store >= 1024 && a.K >= 2048) ? 1 : 2;
  const ConvGemmForce &f = g_conv_gemm_force;
  if (f.path == 4 && f.tile >= 0 && f.tile <= 2) pl.variant = f.tile;
  return true;
}

const char *conv_gemm_v2_name(int dt, const V2Plan &pl) {
  static const char *n[2][3] = {{"conv_gemm_v2<f32,128x128>", "conv_gemm_v2<f32,128x64>", "conv_gemm_v2<f32,64x64>"},
                                {"conv_gemm_v2<bf16,128x128>", "conv_gemm_v2<bf16,128x64>", "conv_gemm_v2<bf16,64x64>"}};
  return label_for_dtype(dt, n[dt == F32 ? 0 : 1][pl.variant]);
}

hipError_t launch_conv_gemm_v2(int dt, const ConvGemmArgs &a, const V2Plan &pl, hipStream_t s) {
  if (dt == F32) {
    switch (pl.variant) {
      case 0: return launch_v2_g<float, 128, 128>(a, pl, s);
      case 1: return launch_v2_g<float, 128, 64>(a, pl, s);
      default: return launch_v2_g<float, 64, 64>(a, pl, s);
    }
  }
  if (dt == F16) {
    switch (pl.variant) {
      case 0: return launch_v2_g<f16, 128, 128>(a, pl, s);
      case 1: return launch_v2_g<f16, 128, 64>(a, pl, s);
      default: return launch_v2_g<f16, 64, 64>(a, pl, s);
    }
  }
  switch (pl.variant) {
    case 0: return launch_v2_g<bf16, 128, 128>(a, pl, s);
    case 1: return launch_v2_g<bf16, 128, 64>(a, pl, s);
    default: return launch_v2_g<bf16, 64, 64>(a, pl, s);
  }
}

}  // namespace sf
