// Macro-tile MFMA implicit GEMM ("mt") for LONG activations in the 16-bit types: the MFMA-bound regime of the path
// (VideoOnsetNet at any batch, the U-Net's deep levels at the guidance batch of BASELINE configs[2]).
//
// Why another GEMM: a CU's L2 -> LDS fill rate (~29 B/clk, MI355X_MICROARCH.md) caps a bm x bn tile at
// 4069 (bm + bn) / (bm bn) B/clk of operand traffic per unit of matrix-core time -- 127 B/clk for 64x64 (ceiling 23 % of
// the dense bf16 peak), 64 B/clk for 128x128, 43 B/clk for 256x128.  conv_gemm_v2 (64x64 ... 128x128 tiles, register
// staging, one LDS buffer) therefore stalls at 230-490 TFLOP/s.  This kernel follows the guide's deep-pipeline recipe
// (cdna_hip_programming.md section 5, "Pipelining across barriers" / the 8-phase template's rules):
//   * 256 x 128 block tile, 512 threads = 8 waves as 4 (rows) x 2 (columns), wave tile 64 x 64 = 2 x 2 MFMA 32x32x16 tiles;
//   * operands go global -> LDS DIRECTLY (`buffer_load_dwordx4 ... lds`, 16 B per lane, out-of-range offsets return zero, so
//     padding rows / taps outside the clip / M and N tails need no predication) into a 3-slot LDS ring of 64-deep K steps;
//     the LDS image is lane-linear, so the bank-conflict swizzle sits on the SOURCE side (16-byte chunk c of row r is fetched
//     from chunk c ^ ((r >> 1) & 7), the key that is conflict-free for the lane groups of ds_read_b128, see the kernel) and on the fragment reads (rule 21 of the guide);
//   * one raw s_barrier per K step and a COUNTED s_waitcnt vmcnt(N): the loads of step k+1 stay in flight across the barrier
//     while step k is multiplied, those of step k+2 are issued right after the barrier (the slot they overwrite was read during
//     step k-1, which every wave has finished when it passes the barrier);
//   * no ordinary global load inside the K loop (hipcc would drain the DMA queue for it): epilogue operands are read after it;
//   * epilogue through LDS: row-major 16-byte stores; bias / per-clip scale / residual / per-clip add / activation as in
//     ConvGemmArgs.
//   * row-LayerNorm fusion at long activations (the guidance batch: 20 ln_modulate launches per branch and step disappear):
//     rowpart_out -- the epilogue also writes (mean, M2) of the STORED values per row and 32-column tile (four lanes hold a row's 32
//     columns of a pass: two quad permutes); ln_colsum + ln_part -- the operand rows go through the matrix cores RAW and the
//     LayerNorm lands on the accumulator, rstd_m * (acc - mean_m * colsum[n]); the row statistics are pooled from the producer's
//     partials, whose loads are issued before the K loop and reduced after it into the tail of the retired ring.
// Geometry: 1-D (taps, stride, nearest upsampling) and video (kt x kh x kw taps), channel counts that are multiples of 64,
// no prologue.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int BK = 64;                 // K step of the 16-bit types: 64 elements = 128 bytes = 8 sixteen-byte chunks per row
constexpr int ROWB = 128;              // bytes per staged row (fp32: 32 elements per K step)
constexpr unsigned OOB = 0x80000000u;
constexpr int NSTAGE = 3;

typedef __attribute__((address_space(3))) void lds_void;

// -DSF_MT_STAMPS: workgroup 0 accumulates, per wave, the shader-clock cycles (s_memtime) its K loop spends in each part of a K step --
// counted wait | barrier | DMA issue | fragment reads + MFMA issue -- plus prologue and epilogue, into g_mt_stamps[wave][6]
// (tools/mt_stamps.hip reads them back).  Off in the product build.
#ifdef SF_MT_STAMPS
__device__ unsigned long long g_mt_stamps[8][8];
#define SF_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#endif

// CAT: K = taps * cin + cin2, the last cin2 columns read row m of a second source (InjectChannels: Conv1x1 over cat[x, ctx])
// WM x WN = 8 waves; a wave owns (BM / WM) x (BN / WN) = (32 TM) x (32 TN) of the block tile
// LNE: the accumulator-side LayerNorm is compiled in (its pooled partials stay in registers across the K loop: only the tiles the
// LayerNorm-folded projections take carry it)
// PRE: the epilogue's residual rows and per-column / per-clip vectors are requested BEFORE the K loop and wait in registers (they are
// older than every DMA piece: the loop's counted waits cover them) -- otherwise every workgroup of a launch ends on one exposed
// global-load round trip, which is 20-50 % of a short-reduction launch.  Single-pass epilogues with a lane's column octet fixed (64 % OCT == 0).
// X3 (T = float): the weights are the split-fp16 image (ConvGemmArgs::wx, the same 128 bytes per row and K step: 32 hi | 32 lo'), the fp32
// activation fragments are split in registers after the LDS read, three v_mfma_f32_32x32x16_f16 per product into two accumulators
// (common.h, x3_split): fp32-grade results at 16-bit matrix rates.  DMA ring, swizzle and epilogue are those of the fp32 instantiation.
template <typename T, int BM, int BN, int WM, int WN, int GEOM, bool CAT, int NST = NSTAGE, int EP = 1, bool LNE = false, bool PRE = false, int X3 = 0, bool AX = false>
__global__ __launch_bounds__(512) void conv_gemm_mt_kernel(const ConvGemmArgs a, const int mtiles, const int ntiles, const unsigned bytesA,
                                                           const unsigned bytesA2, const unsigned bytesW) {
  constexpr int ES = sizeof(T);   // fp32 (training, the parity engine's long activations): same byte geometry, v_mfma_f32_32x32x2_f32
  static_assert(!X3 || (sizeof(T) == 4 && !LNE && !PRE), "split mode: fp32 activations");
  static_assert(WM * WN == 8 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % 64 == 0 && BN % 64 == 0, "tile / wave grid mismatch");
  constexpr int RM = BM / WM, RN = BN / WN;          // rows / columns of a wave's tile
  constexpr int TM = RM / 32, TN = RN / 32;
  constexpr int STAGE = (BM + BN) * ROWB;          // bytes per ring slot: A rows then W rows
  constexpr int PA = BM / 64, PB = BN / 64;         // DMA instructions per thread and K step (8 rows per wave-instruction, 8 waves)
  constexpr int NLD = PA + PB;
  using frag = typename std::conditional<sizeof(T) == 4, f32x4, typename Frag16<T>::type>::type;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- block -> tile: each XCD gets a contiguous run of the m-major tile list (the column tiles of a row band share its A panel)
  int wg;
  {
    const int nwg = mtiles * ntiles, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = wg / ntiles, nt = wg - mt * ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(X3 ? a.wx : a.w), 0, bytesW, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(CAT ? a.src2 : a.src), 0, CAT ? bytesA2 : 0, 0x00020000);

  // ---- DMA lane geometry: a wave-instruction fills 8 rows x 128 B; lane -> (row lane>>3, LDS chunk lane&7), source chunk swizzled
  const int lrow = lane >> 3;
  // Swizzle key of a staged row = (row >> 1) & 7.  A ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
  // (+ 32 for the upper half-wave), 16 lanes per LDS cycle over 64 banks of 4 bytes: a fragment read takes 16 rows whose bank quad is
  // 8 (row & 1) + chunk, so the eight even (odd) rows of a group need eight different chunks -- (row >> 1) & 7 gives exactly that for both
  // groups, where the row & 7 used until round 3 put two rows on every quad (SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE,
  // profiles/r3_l_cfg2_lds.csv).  A DMA piece is rows (i * 8 + wave) * 8 + lrow, i.e. key (4 wave + (lrow >> 1)) & 7.
  const unsigned gchunk_b = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lrow >> 1)) & 7)) * 16);
  int rbase[PA], rp0[PA], rh[PA], rw_[PA];
  unsigned vmask[PA], woff[PB], roff2[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + (i * 8 + wave) * 8 + lrow;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    vmask[i] = vm ? 0u : OOB;
    roff2[i] = CAT ? (((unsigned)(mm * a.src2_ld * ES) + gchunk_b) | vmask[i]) : OOB;
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
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int n = n0 + (j * 8 + wave) * 8 + lrow;
    woff[j] = n < a.N ? ((unsigned)(n * a.K * ES) + gchunk_b) : OOB;
  }
  const int pmax = (a.Lsrc << a.up_shift) - 1;
  unsigned cur[PA];
  auto retap = [&](int t) {
    if constexpr (GEOM == 0) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int p = rp0[i] + t;
        const unsigned bad = ((unsigned)p > (unsigned)pmax) ? OOB : 0u;
        cur[i] = ((unsigned)(((rbase[i] + (max(p, 0) >> a.up_shift)) * a.src_ld) * ES) + gchunk_b) | bad | vmask[i];
      }
    } else {
      const int dw = t % a.kw;
      const int r = t / a.kw;
      const int dh = r % a.kh, dt = r / a.kh;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int ti = rp0[i] + dt, hi = rh[i] + dh, wi = rw_[i] + dw;
        const bool ok = (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
        cur[i] = ((unsigned)(((((rbase[i] + max(ti, 0)) * a.Hi + max(hi, 0)) * a.Wi + max(wi, 0)) * a.src_ld) * ES) + gchunk_b) |
                 (ok ? 0u : OOB) | vmask[i];
      }
    }
  };

  // ---- load stream state (runs two K steps ahead of the multiply) ------------------------------------------------
  const int nk = a.K / (ROWB / ES);
  const unsigned tap_bytes = (unsigned)(a.cin * ES);
  unsigned cb = 0, kb = 0;   // byte offset inside the tap's channels / inside a W row
  int tap = 0;
  bool second = false;   // the stream has reached the concatenated source
  retap(0);
  auto issue = [&](int slot) {
    unsigned char *base = smem + slot * STAGE;
    const __amdgpu_buffer_rsrc_t rs = (CAT && second) ? rA2 : rA;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(base + (i * 8 + wave) * 1024), 16, (int)(cur[i] + cb), 0, 0, 0);
#pragma unroll
    for (int j = 0; j < PB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void *)(base + BM * ROWB + (j * 8 + wave) * 1024), 16, (int)(woff[j] + kb), 0, 0, 0);
    kb += ROWB;
    cb += ROWB;
    if (!second && cb >= tap_bytes) {   // wave-uniform
      cb = 0;
      ++tap;
      if (tap < a.taps) retap(tap);
      else if (CAT) {
        second = true;
#pragma unroll
        for (int i = 0; i < PA; ++i) cur[i] = roff2[i];
      }
    }
  };

  f32x16 acc[TM][TN];
  f32x16 accL[X3 == X3_F16 ? TM : 1][X3 == X3_F16 ? TN : 1];   // fp16 split: the cross terms hi lo' + lo' hi (scaled by 2048); bf16 split: one accumulator
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (X3 == X3_F16) accL[i][j][r] = 0.f;
      }

  // fragment read offsets inside a slot: row-dependent part once, the k sub-step enters through the XOR
  unsigned offA[TM], offB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * RM + i * 32 + fr;
    offA[i] = (unsigned)(row * ROWB);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * RN + j * 32 + fr;
    offB[j] = (unsigned)(BM * ROWB + row * ROWB);
  }
  const unsigned sw = (unsigned)((fr >> 1) & 7);   // rows of a fragment: every row base is a multiple of 32, so (row >> 1) & 7 == (fr >> 1) & 7

  // ---- PRE: epilogue operands of this lane's (row, column octet) pairs ---------------------------------------------------------------
  constexpr int EIT = (RM / EP) * (RN / 8) / 64;      // epilogue iterations per lane and pass
  static_assert(!PRE || (EP == 1 && sizeof(T) == 2 && GEOM == 0 && 64 % (RN / 8) == 0), "PRE: single-pass 16-bit 1-D epilogues");
  Vec16<T> pre_res[PRE ? EIT : 1];
  f32x4 pre_bias[2], pre_bs[PRE ? EIT : 1][2], pre_ba[PRE ? EIT : 1][2];
  if constexpr (PRE) {
    constexpr int OCTP = RN / 8;
    const int octp = lane % OCTP;
    const int n = n0 + wn * RN + octp * 8;
    const bool full = n + 8 <= a.N;
    const int nc = full ? n : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) pre_bias[h] = (a.bias && full) ? *reinterpret_cast<const f32x4 *>(a.bias + nc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int rl = (it * 64 + lane) / OCTP;
      const int m = m0 + wm * RM + rl;
      const bool live = m < a.M && n < a.n_store;
      const int mc = min(m, a.M - 1);
      const int b = mc / a.Lout;
      pre_res[it] = (a.res && live) ? ld16<T>(static_cast<const T *>(a.res) + (size_t)mc * a.res_ld + n) : zero16<T>();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pre_bs[it][h] = (a.bscale && full) ? *reinterpret_cast<const f32x4 *>(a.bscale + (size_t)b * a.bscale_ld + nc + 4 * h) : f32x4{1.f, 1.f, 1.f, 1.f};
        pre_ba[it][h] = (a.badd && full) ? *reinterpret_cast<const f32x4 *>(a.badd + (size_t)b * a.badd_ld + nc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }

  // ---- LayerNorm on the accumulator: this thread's share of its row's producer partials, requested ahead of the DMA stream (they
  // are older than every DMA piece, so the counted waits of the K loop cover them; first use is after the loop) ----------------------
  static_assert(!LNE || (ES == 2 && GEOM == 0 && !CAT && (BM == 128 || BM == 256)), "accumulator-side LayerNorm: 16-bit 1-D projections");
  constexpr int TPR = LNE ? 512 / BM : 4;             // threads per row of the block tile (2 or 4: one DPP quad)
  constexpr int LNP = LNE ? 32 / TPR : 1;             // partials per thread (ln_nt <= 32)
  const bool ln_epi = LNE && a.ln_colsum != nullptr;
  float2 lnp[LNP];
  if (ln_epi) {
    const int m = min(m0 + tid / TPR, a.M - 1), t0_ = tid % TPR;
#pragma unroll
    for (int j = 0; j < LNP; ++j) {
      const int pidx = t0_ + TPR * j;
      lnp[j] = pidx < a.ln_nt ? *reinterpret_cast<const float2 *>(a.ln_part + ((size_t)m * a.ln_nt + pidx) * 2) : make_float2(0.f, 0.f);
    }
  }

  // ---- prologue: two K steps in flight ---------------------------------------------------------------------------
#ifdef SF_MT_STAMPS
  SF_STAMP(t_begin);
  unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_mma = 0;
#endif
  issue(0);
  if (NST == 3 && nk > 1) issue(1);
#ifdef SF_MT_STAMPS
  SF_STAMP(t_loop);
#endif

  for (int k = 0; k < nk; ++k) {
#ifdef SF_MT_STAMPS
    SF_STAMP(t0);
#endif
    if constexpr (NST == 3) {
      // own DMA of step k has landed when at most the NLD loads of step k+1 are still outstanding
      if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SF_MT_STAMPS
      SF_STAMP(t1);
#endif
      __builtin_amdgcn_s_barrier();   // every wave's step-k data is in LDS; every wave is done reading step k-1's slot
#ifdef SF_MT_STAMPS
      SF_STAMP(t2);
#endif
      if (k + 2 < nk) issue((k + 2) % 3);
#ifdef SF_MT_STAMPS
      SF_STAMP(t3);
      c_wait += t1 - t0;
      c_bar += t2 - t1;
      c_issue += t3 - t2;
#endif
    } else {
      // two slots (half the LDS: two workgroups share a CU and cover each other's prologue / epilogue): one step ahead only
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SF_MT_STAMPS
      SF_STAMP(t1);
#endif
      __builtin_amdgcn_s_barrier();
#ifdef SF_MT_STAMPS
      SF_STAMP(t2);
#endif
      if (k + 1 < nk) issue((k + 1) % 2);
#ifdef SF_MT_STAMPS
      SF_STAMP(t3);
      c_wait += t1 - t0;
      c_bar += t2 - t1;
      c_issue += t3 - t2;
#endif
    }
#ifdef SF_MT_STAMPS
    SF_STAMP(t4);
#endif
    const unsigned char *slot = smem + (k % NST) * STAGE;
    if constexpr (X3) {
      // a K step = 32 k = two 16-deep products.  Activation row: 8 chunks of 4 floats, the half-wave fh takes chunks 4 s + 2 fh, + 1
      // (k = 16 s + 8 fh + 0..7); weight row: chunks 0-3 = hi, 4-7 = lo', chunk 2 s + fh holds the same eight k.
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        using xv = typename X3P<X3 ? X3 : 1>::v8;
        xv ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (AX) {   // the producer wrote the rows already split (ConvGemmArgs::src_x3): same chunk geometry as the weights
            ah[i] = *reinterpret_cast<const xv *>(slot + offA[i] + (unsigned)(((2 * s2 + fh) ^ sw) * 16));
            al[i] = *reinterpret_cast<const xv *>(slot + offA[i] + (unsigned)(((4 + 2 * s2 + fh) ^ sw) * 16));
          } else {
            const f32x4 p = *reinterpret_cast<const f32x4 *>(slot + offA[i] + (unsigned)(((4 * s2 + 2 * fh) ^ sw) * 16));
            const f32x4 q = *reinterpret_cast<const f32x4 *>(slot + offA[i] + (unsigned)(((4 * s2 + 2 * fh + 1) ^ sw) * 16));
            x3_split<X3 ? X3 : 1>(p, q, ah[i], al[i]);
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const xv *>(slot + offB[j] + (unsigned)(((2 * s2 + fh) ^ sw) * 16));
          bl[j] = *reinterpret_cast<const xv *>(slot + offB[j] + (unsigned)(((4 + 2 * s2 + fh) ^ sw) * 16));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (X3 == X3_BF16) x3_mfma1_bf16(ah[i], al[i], bh[j], bl[j], acc[i][j]);
            else x3_mfma<X3_F16>(ah[i], al[i], bh[j], bl[j], acc[i][j], accL[i][j]);
          }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned ch = (unsigned)(((2 * ks + fh) ^ sw) * 16);
      frag af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const frag *>(slot + offA[i] + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const frag *>(slot + offB[j] + ch);
      if constexpr (ES == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32x16(af[i], bf[j], acc[i][j]);
      } else {
        // fp32: the half-wave fh holds floats 4 (2 ks + fh) + e of the K step; element e of both halves forms one k pair of a 32x32x2
        // product (the k order inside a step is permuted identically for both operands)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
    }
    }   // !X3
#ifdef SF_MT_STAMPS
    asm volatile("s_nop 0" ::"v"(acc[0][0][0]));   // the last MFMA of the step has written its accumulator
    SF_STAMP(t5);
    c_mma += t5 - t4;
#endif
  }
#ifdef SF_MT_STAMPS
  SF_STAMP(t_loop_end);
#endif

  // ---- epilogue through LDS: each wave parks its RM x RN fp32 tile (EP = 2: half of the rows of every 32-row MFMA tile at a time, so
  // that the parking area fits inside a two-slot ring), then streams it out row-major ----------------------------------------------
  __builtin_amdgcn_s_barrier();   // all fragment reads of the last steps are done before the ring is reused
  constexpr int LDR = RN + 4;
  constexpr int RPT = 32 / EP;     // rows of a 32-row MFMA tile handled per pass
  constexpr int RMP = RM / EP;     // rows a wave parks per pass
  float *red = reinterpret_cast<float *>(smem) + (size_t)wave * RMP * LDR;
  float *rowstat = reinterpret_cast<float *>(smem) + (size_t)8 * RMP * LDR;   // (mean, rstd) per row of the block tile, behind the parking area
  if (ln_epi) {
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < LNP; ++j) sm += lnp[j].x;          // absent partials are zeros
    sm += dpp_mov_f<0xb1, 0xf>(sm);
    if constexpr (TPR == 4) sm += dpp_mov_f<0x4e, 0xf>(sm);
    const float mean = sm / (float)a.ln_nt;                // every partial covers 32 channels
    float dq = 0.f;
#pragma unroll
    for (int j = 0; j < LNP; ++j) {
      if ((tid % TPR) + TPR * j < a.ln_nt) {
        const float d = lnp[j].x - mean;
        dq += fmaf(32.f * d, d, lnp[j].y);
      }
    }
    dq += dpp_mov_f<0xb1, 0xf>(dq);
    if constexpr (TPR == 4) dq += dpp_mov_f<0x4e, 0xf>(dq);
    if (tid % TPR == 0) {
      rowstat[2 * (tid / TPR)] = mean;
      rowstat[2 * (tid / TPR) + 1] = rsqrtf(dq / (float)a.cin + a.ln_eps);
    }
    __syncthreads();
  }
  T *out = static_cast<T *>(a.out);
  const T *res = static_cast<const T *>(a.res);
  const bool has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
  constexpr int OCT = RN / 8;                      // 8-column groups per tile row
#pragma unroll
  for (int pass = 0; pass < EP; ++pass) {
  if (pass) __builtin_amdgcn_wave_barrier();       // the previous pass's reads precede these writes (same-wave LDS order)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (EP == 2 && (r >> 3) != pass) continue;   // registers 0-7 = rows 0-15 of the tile, 8-15 = rows 16-31
        const int lr = EP == 1 ? (r & 3) + 8 * (r >> 2) + 4 * fh : (r & 3) + 8 * ((r >> 2) & 1) + 4 * fh;
        if constexpr (X3 == X3_F16) red[(i * RPT + lr) * LDR + j * 32 + fr] = fmaf(accL[i][j][r], X3P<X3_F16>::INV, acc[i][j][r]);
        else red[(i * RPT + lr) * LDR + j * 32 + fr] = acc[i][j][r];
      }
  __builtin_amdgcn_wave_barrier();   // same-wave LDS operations execute in order; this only pins the compiler
#pragma unroll
  for (int it = 0; it < RMP * OCT / 64; ++it) {
    const int idx = it * 64 + lane;
    const int rl = idx / OCT, oct = idx - rl * OCT;
    const int m = m0 + wm * RM + (rl / RPT) * 32 + pass * RPT + (rl % RPT), n = n0 + wn * RN + oct * 8;
    const bool live = m < a.M && n < a.n_store;
    const int mc = min(m, a.M - 1);
    const bool full = n + 8 <= a.N;            // whole octet inside the real columns (pad columns [N, n_store) are stored as zeros)
    const int nc = full ? n : 0;
    const int b = (has_bs || has_ba) ? mc / a.Lout : 0;
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(red + rl * LDR + oct * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(red + rl * LDR + oct * 8 + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (ln_epi) {   // LayerNorm of the raw operand rows, folded into the accumulator (colsum[n] = sum_k w[n][k])
      const int rb = wm * RM + (rl / RPT) * 32 + pass * RPT + (rl % RPT);   // row inside the block tile
      const float mu = rowstat[2 * rb], rstd = rowstat[2 * rb + 1];
      const int ncs = min(n, a.N - 8);   // (ln_epi implies N % 8 == 0: see conv_gemm_mt_ok)
      const f32x4 c0 = *reinterpret_cast<const f32x4 *>(a.ln_colsum + ncs), c1 = *reinterpret_cast<const f32x4 *>(a.ln_colsum + ncs + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = rstd * (v[e] - mu * c0[e]);
        v[4 + e] = rstd * (v[4 + e] - mu * c1[e]);
      }
    }
    float rres[8];   // residual rows are n_store wide (pad columns hold zeros)
#pragma unroll
    for (int e = 0; e < 8; ++e) rres[e] = 0.f;
    if constexpr (PRE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) rres[e] = pre_res[it].get(e);
    } else if (res && live) {
      if constexpr (ES == 2) {
        const Vec16<T> rv = ld16<T>(res + (size_t)mc * a.res_ld + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) rres[e] = rv.get(e);
      } else {
        const Vec16<T> r0 = ld16<T>(res + (size_t)mc * a.res_ld + n), r1 = ld16<T>(res + (size_t)mc * a.res_ld + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rres[e] = r0.get(e);
          rres[4 + e] = r1.get(e);
        }
      }
    }
    float bi[8], sc[8], ad[8];
    if (PRE && full) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bi[4 * h + e] = pre_bias[h][e];
          sc[4 * h + e] = pre_bs[it][h][e];
          ad[4 * h + e] = pre_ba[it][h][e];
        }
    } else if (full) {   // 16-byte vectors of the per-column operands (n is a multiple of 8)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + nc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 sv = has_bs ? *reinterpret_cast<const f32x4 *>(a.bscale + (size_t)b * a.bscale_ld + nc + 4 * h) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 av = has_ba ? *reinterpret_cast<const f32x4 *>(a.badd + (size_t)b * a.badd_ld + nc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bi[4 * h + e] = bv[e];
          sc[4 * h + e] = sv[e];
          ad[4 * h + e] = av[e];
        }
      }
    } else {      // the octet that straddles N (odd channel counts of the onset net): element-wise, clamped
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ne = min(n + e, a.N - 1);
        bi[e] = a.bias ? a.bias[ne] : 0.f;
        sc[e] = has_bs ? a.bscale[(size_t)b * a.bscale_ld + ne] : 1.f;
        ad[e] = has_ba ? a.badd[(size_t)b * a.badd_ld + ne] : 0.f;
      }
    }
    float xo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = (v[e] + bi[e]) * sc[e] + rres[e] + ad[e];
      if (a.act == 1) x = fmaxf(x, 0.f);
      xo[e] = (n + e < a.N) ? x : 0.f;
    }
    if (live) {
      if constexpr (ES == 2) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, xo[e]);
        st16<T>(out + (size_t)m * a.out_ld + n, o);
        if (a.rowpart_out) {   // (mean, M2) of the row's 32 stored values of this column tile: the four lanes of a quad hold them
#pragma unroll
          for (int e = 0; e < 8; ++e) xo[e] = o.get(e);
        }
      } else {
        Vec16<T> o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o0.set(e, xo[e]);
          o1.set(e, xo[4 + e]);
        }
        st16<T>(out + (size_t)m * a.out_ld + n, o0);
        st16<T>(out + (size_t)m * a.out_ld + n + 4, o1);
      }
    }
    if constexpr (ES == 2 && GEOM == 0) {
      if (a.rowpart_out) {   // every lane of the wave takes part (rows / columns outside the tensor contribute zeros and store nothing)
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += live ? xo[e] : 0.f;
        sm += dpp_mov_f<0xb1, 0xf>(sm);
        sm += dpp_mov_f<0x4e, 0xf>(sm);
        const float mean = sm * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (live ? xo[e] : 0.f) - mean;
          q = fmaf(d, d, q);
        }
        q += dpp_mov_f<0xb1, 0xf>(q);
        q += dpp_mov_f<0x4e, 0xf>(q);
        if (live && (oct & 3) == 0) *reinterpret_cast<float2 *>(a.rowpart_out + ((size_t)m * a.rowpart_nt + (n >> 5)) * 2) = make_float2(mean, q);
      }
    }
  }
  }
#ifdef SF_MT_STAMPS
  if (blockIdx.x == 0 && lane == 0) {
    SF_STAMP(t_end);
    unsigned long long *o = g_mt_stamps[wave];
    o[0] = c_wait;
    o[1] = c_bar;
    o[2] = c_issue;
    o[3] = c_mma;
    o[4] = t_loop - t_begin;
    o[5] = t_end - t_loop_end;
    o[6] = t_end - t_begin;
    o[7] = (unsigned long long)nk;
  }
#endif
}

template <typename T, int BM, int BN, int WM, int WN, int GEOM, bool CAT, int NST = NSTAGE, int EP = 1, bool LNE = false, bool PRE = false, int X3 = 0, bool AX = false>
hipError_t launch_mt(const ConvGemmArgs &a, hipStream_t s) {
  constexpr size_t ring = (size_t)NST * (BM + BN) * ROWB;
  constexpr size_t redb = (size_t)8 * (BM / WM / EP) * (BN / WN + 4) * sizeof(float) + (size_t)BM * 2 * sizeof(float);   // parking area + rowstat
  constexpr size_t lds = ring > redb ? ring : redb;
  static_assert(lds <= 160 * 1024, "LDS budget");
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.n_store + BN - 1) / BN;
  size_t bA;
  if (GEOM == 0) bA = (size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * sizeof(T);
  else bA = (size_t)((a.M + a.To * a.Ho * a.Wo - 1) / (a.To * a.Ho * a.Wo)) * a.Ti * a.Hi * a.Wi * a.src_ld * sizeof(T);
  const size_t bW = (size_t)a.N * a.K * sizeof(T);
  const size_t bA2 = CAT ? (size_t)a.M * a.src2_ld * sizeof(T) : 0;
  auto kern = conv_gemm_mt_kernel<T, BM, BN, WM, WN, GEOM, CAT, NST, EP, LNE, PRE, X3, AX>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(512), lds, s, a, mtiles, ntiles, (unsigned)bA, (unsigned)bA2, (unsigned)bW);
  return hipGetLastError();
}

// the tile variants the accumulator-side LayerNorm is instantiated for (the LayerNorm-folded qkv projections: 128x192 two-slot, 128x64)
static bool mt_lne_variant(int v) { return v == 6 || v == 7; }

template <typename T, int GEOM, bool CAT> hipError_t launch_mt_v(const ConvGemmArgs &a, int variant, hipStream_t s) {
  if constexpr (GEOM == 0 && !CAT && sizeof(T) == 2) {
    if (a.ln_colsum) {
      if (variant == 6) return launch_mt<T, 128, 192, 4, 2, GEOM, CAT, 2, 2, true>(a, s);
      if (variant == 7) return launch_mt<T, 128, 64, 4, 2, GEOM, CAT, 3, 1, true>(a, s);
      return hipErrorInvalidValue;
    }
  } else {
    if (a.ln_colsum) return hipErrorInvalidValue;
  }
  switch (variant) {
    case 1: return launch_mt<T, 128, 128, 2, 4, GEOM, CAT>(a, s);   // short M: twice the tiles of 256x128
    case 2: return launch_mt<T, 128, 192, 4, 2, GEOM, CAT>(a, s);   // column counts that are multiples of 192 but not of 128
    case 3: return launch_mt<T, 192, 128, 2, 4, GEOM, CAT>(a, s);   // 4/3 of the tiles of 256x128 at 5/6 of its fill per tile
    case 4: return launch_mt<T, 256, 64, 8, 1, GEOM, CAT>(a, s);    // outputs of <= 64 columns
    case 5: return launch_mt<T, 128, 128, 2, 4, GEOM, CAT, 2, 2>(a, s);   // two-slot ring, 64 KB of LDS: two workgroups per CU
    case 6: return launch_mt<T, 128, 192, 4, 2, GEOM, CAT, 2, 2>(a, s);   // the same for 192-wide column tiles (80 KB)
    case 7:                                                               // three-slot ring of a 128x64 tile (72 KB): two workgroups per CU
      if constexpr (GEOM == 0 && sizeof(T) == 2) {
        static const bool no_pre = tune_env("SF_MT_NO_PRE") != nullptr;      // A/B aid: epilogue operands loaded in the epilogue as before
        if (!no_pre) return launch_mt<T, 128, 64, 4, 2, GEOM, CAT, 3, 1, false, true>(a, s);
      }
      return launch_mt<T, 128, 64, 4, 2, GEOM, CAT, 3, 1>(a, s);
    case 8: return launch_mt<T, 192, 128, 2, 4, GEOM, CAT, 2, 2>(a, s);   // two-slot 192x128 (80 KB): two workgroups per CU, 5/6 of the fill of 128x128
    case 9: return launch_mt<T, 256, 64, 8, 1, GEOM, CAT, 2, 2>(a, s);    // two-slot 256x64 (80 KB): the same for outputs of <= 64 columns
    case 10: return launch_mt<T, 256, 256, 2, 4, GEOM, CAT, 2, 2>(a, s);  // two-slot 256x256 (128 KB ring), wave tile 128x64: half the fill and 3/4 of the LDS reads per FLOP of 192x128
    default: return launch_mt<T, 256, 128, 4, 2, GEOM, CAT>(a, s);
  }
}

// split mode (fp32 activations x split-fp16 weights): wave tiles of at least 32 x 64, so that a split activation fragment feeds six MFMAs
template <bool CAT, int MODE, bool AX = false> hipError_t launch_mt_x3(const ConvGemmArgs &a, int v, hipStream_t s) {
  switch (v) {
    case 0: return launch_mt<float, 256, 128, 4, 2, 0, CAT, 3, 1, false, false, MODE, AX>(a, s);   // wave tile 64 x 64, one workgroup per CU
    case 1: return launch_mt<float, 128, 128, 4, 2, 0, CAT, 3, 1, false, false, MODE, AX>(a, s);   // wave tile 32 x 64, three slots (96 KB)
    case 5: return launch_mt<float, 128, 128, 4, 2, 0, CAT, 2, 2, false, false, MODE, AX>(a, s);   // the same with two slots (64 KB): two workgroups per CU
    case 7: return launch_mt<float, 128, 64, 4, 2, 0, CAT, 3, 1, false, false, MODE, AX>(a, s);    // wave tile 32 x 32 (72 KB): two workgroups per CU
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

int conv_gemm_mt_variant(const ConvGemmArgs &a);
// eligibility (what the kernel implements) -- the CHOICE between this kernel and conv_gemm_v2 is conv_gemm_prefers_mt
bool conv_gemm_mt_ok(int dt, const ConvGemmArgs &a) {
  const size_t es = dt == F32 ? 4 : 2;
  const int bke = (int)(ROWB / es), vec = (int)(16 / es);   // elements per K step / per 16-byte access
  if (dt == F32 && a.geom != 0) return false;                // fp32: the U-Net's 1-D geometry only (training, parity engine)
  if (a.pro != 0 || (a.cin % bke) || (a.cin2 % bke) || a.taps < 1 || a.K != a.taps * a.cin + a.cin2) return false;
  if (a.wx && a.wx_mode == X3_BF16 && a.cin2) return false;
  if (a.src_x3 && (dt != F32 || !a.wx || a.wx_mode != X3_F16 || a.cin2 || a.geom != 0)) return false;
  if (a.cin2 && (a.geom != 0 || !a.src2 || (a.src2_ld % vec) || a.src2_ld < a.cin2 || (size_t)a.M * a.src2_ld * es >= 0x7FFFFFF0ull)) return false;
  if (a.out_f32 || a.act > 1 || a.ln_ss || a.res_ln) return false;
  if (a.rowpart_out && (dt == F32 || a.geom != 0 || (a.n_store % 32) || a.rowpart_nt * 32 != a.n_store || a.N != a.n_store)) return false;
  if ((a.ln_part != nullptr) != (a.ln_colsum != nullptr)) return false;   // accumulator-side LayerNorm only (no operand transform on a DMA ring)
  if (a.ln_colsum && (dt == F32 || a.geom != 0 || a.taps != 1 || a.cin2 || a.stride != 1 || a.up_shift || a.Lout != a.Lsrc || a.ln_nt * 32 != a.cin ||
                      a.ln_nt > 32 || (a.N % 8) || (reinterpret_cast<uintptr_t>(a.ln_colsum) & 15) || !mt_lne_variant(conv_gemm_mt_variant(a))))
    return false;
  if ((a.bscale && (a.bscale_ld % 4)) || (a.badd && (a.badd_ld % 4))) return false;
  if ((a.n_store % 8) || a.n_store < a.N || (a.out_ld % 8) || a.out_ld < a.n_store || (a.res && ((a.res_ld % 8) || a.res_ld < a.n_store)) || (a.src_ld % vec)) return false;
  if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
  if (a.geom == 1 && (a.bscale || a.badd)) return false;
  const size_t lim = 0x7FFFFFF0ull;
  size_t bA;
  if (a.geom == 0) bA = (size_t)(a.M / a.Lout + 1) * a.Lsrc * a.src_ld * es;
  else bA = (size_t)(a.M / (a.To * a.Ho * a.Wo) + 1) * a.Ti * a.Hi * a.Wi * a.src_ld * es;
  if (bA >= lim || (size_t)a.N * a.K * es >= lim) return false;
  return true;
}

// fp32 choice (training step, the fp32 engine's long activations): the fp32 matrix pipe needs 64 cycles per 32x32x2 product, so a K step
// of ANY tile is MFMA time (2048 cycles per SIMD against ~400 of wait + barrier + DMA issue) and the tile only has to fill the chip:
// 128x64 (three slots, two workgroups per CU) below 512 tiles of 128x128, 128x128 (two slots, two per CU) above.  Alone on the chip
// (tools/gemm_f32.py, batch 4 x 2^18 samples): 16384 x 128 x 384 25.3 -> 19.3 us, 8192 x 256 x 768 39.3 -> 32.9, 4096 x 512 x 1536 66.5 -> 60.1,
// 8192 x 1536 x 256 137.7 -> 62.0, 4096 x 1536 x 512 126.2 -> 58.6 us (110 TFLOP/s of the 157 fp32 peak).
bool conv_gemm_prefers_mt_f32(const ConvGemmArgs &a) {
  static const long min_tiles = [] {   // tuning hook: SF_MT_F32_TILES=0 keeps fp32 off the macro tiles
    const char *e = tune_env("SF_MT_F32_TILES");
    return e ? atol(e) : 256L;   // a full round of 128x64 tiles: at 128 tiles the wave-split-K / 64x64 kernels win (41 vs 57 us, 71 vs 110 us)
  }();
  if (min_tiles <= 0 || a.geom != 0 || a.K < 256) return false;
  return (long)((a.M + 127) / 128) * ((a.n_store + 63) / 64) >= min_tiles;
}
// split mode: the matrix pipe is fast again, so the choice follows the 16-bit rule (launches that give most CUs a tile)
static bool conv_gemm_prefers_mt_x3(const ConvGemmArgs &a) {
  static const long min_tiles = [] {   // tuning hook: fewest 128x64 tiles that take the macro tiles (0 = never)
    const char *e = tune_env("SF_MT_X3_TILES");
    return e ? atol(e) : 128L;
  }();
  if (min_tiles <= 0 || a.geom != 0 || a.K < 256) return false;
  return (long)((a.M + 127) / 128) * ((a.n_store + 63) / 64) >= min_tiles;
}
static int conv_gemm_mt_x3_variant(const ConvGemmArgs &a) {
  static const int forced = [] { const char *e = tune_env("SF_MT_X3_VARIANT"); return e ? atoi(e) : -1; }();   // tuning hook: 0, 1, 5, 7
  if (forced >= 0) return forced;
  // 128x128 with two slots (two workgroups per CU, wave tile 32 x 64: a split activation fragment feeds six MFMAs) wherever the column
  // count allows it: same-box A/B on configs[2], 128x128 everywhere 93.5 vs 91.0 steps/s with 128x64 below 256 tiles (profiles/r6_c_ab_x3.txt);
  // 256x128 from two full rounds of them
  const long t256 = (long)((a.M + 255) / 256) * ((a.n_store + 127) / 128);
  if (a.n_store <= 64 || ((a.n_store % 128) && (a.n_store % 128) <= 64 && a.n_store % 64 == 0 && a.n_store <= 192)) return 7;
  if (t256 >= 512) return 0;
  // alone on the chip (training: one stream) a launch of fewer than 256 tiles of 128x128 leaves CUs without work: 128x64 tiles there
  // (training step, same box: 30.9 vs 35.8 us per launch, profiles/r6_e_train_kernel_stats.csv vs r6_f)
  if (a.solo && (long)((a.M + 127) / 128) * ((a.n_store + 127) / 128) < 256 && a.n_store % 64 == 0) return 7;
  return 5;
}
bool conv_gemm_src_x3_ok(const ConvGemmArgs &a) {
  ConvGemmArgs p = a;
  p.src_x3 = 0;
  return a.wx && a.wx_mode == X3_F16 && !a.cin2 && a.geom == 0 && conv_gemm_mt_wanted(F32, p);
}
bool conv_gemm_mt_wanted(int dt, const ConvGemmArgs &a) {
  if (!conv_gemm_mt_ok(dt, a)) return false;
  if (dt == F32 && a.wx) return conv_gemm_prefers_mt_x3(a);
  return dt == F32 ? conv_gemm_prefers_mt_f32(a) : conv_gemm_prefers_mt(a);
}

// tile variant: 0 = 256x128, 1 = 128x128, 2 = 128x192, 3 = 192x128, 4 = 256x64 (three-slot ring, one workgroup per CU);
//               5 = 128x128, 6 = 128x192, 8 = 192x128, 9 = 256x64 with a two-slot ring and a two-pass epilogue (two workgroups per CU);
//               7 = 128x64 with three slots (two workgroups per CU)
int conv_gemm_mt_variant(const ConvGemmArgs &a) {
  static const int forced = [] {   // tuning hook
    const char *e = tune_env("SF_MT_VARIANT");
    return e ? atoi(e) : -1;
  }();
  if (forced >= 0 && forced <= 10) return forced;
  auto cols = [&](int bn) { return (long)((a.n_store + bn - 1) / bn) * bn; };
  static const int rule = [] {   // tuning hook: 0 = three-slot rings only, 1 = two-slot rings for every geometry, default: video geometry only
    const char *e = tune_env("SF_MT_RULE");
    return e ? atoi(e) : -1;
  }();
  // 192-wide tiles: column counts they cover without empty tiles (192, 576, 960), and short reductions on counts both tile
  // exactly (the 1536-column qkv projections)
  const bool wide = cols(192) < cols(128) || (a.n_store % 192 == 0 && a.K <= 1024);
  const long t256 = (long)((a.M + 255) / 256) * ((a.n_store + 127) / 128);
  if (rule == 1 || (rule < 0 && a.geom == 1)) {
    // Two-slot rings (64 / 80 KB of LDS) put TWO workgroups on a CU: one's prologue and epilogue hide under the other's K loop.
    // The onset net's convolutions stream activations that no cache holds (3 M rows at N = 32) through short reductions
    // (K = 432 ... 2304 for most of its FLOPs): 477 -> 570 TFLOP/s.  Alone on the chip the two-slot 128x128 tile also wins 11 of
    // 12 U-Net guidance-batch shapes (tools/mt_variants.py, up to 30 % on the K = 256-512 projections), but inside the U-Net
    // step, where every GEMM starts on cold operands, the deeper three-slot ring holds its own (154 vs 162 steps/s at batch 32
    // with guidance), so the 1-D geometry keeps the rule below.
    // Taller tiles with the same two-slot ring (80 KB, still two workgroups per CU) move 5/6 of the bytes per FLOP through the
    // L2 -> LDS fill: 192x128 for 128x128 (+2.8 % on the whole net), 256x64 for 128x64 (+1 %), profiles/r3_j_onset_variants.txt --
    // where the launch still has two rounds of them (the 7x7 and 14x14 stages keep the smaller tiles).
    static const int v_thin = [] { const char *e = tune_env("SF_MT_VIDEO_THIN"); return e ? atoi(e) : 9; }();   // tuning hooks
    static const int v_sq = [] { const char *e = tune_env("SF_MT_VIDEO_SQ"); return e ? atoi(e) : 8; }();
    static const long tall_min = [] { const char *e = tune_env("SF_MT_VIDEO_TALL_MIN"); return e ? atol(e) : 1024L; }();
    if (a.n_store <= 64) return (long)((a.M + 255) / 256) >= tall_min ? v_thin : 7;   // 128x64 tiles (three slots still fit twice)
    if (wide) return 6;
    return (long)((a.M + 191) / 192) * ((a.n_store + 127) / 128) >= tall_min ? v_sq : 5;
  }
  {
    static const int wide_small = [] {   // tuning hook: 192-wide candidates with at most this many 128x128 tiles take the 128x64 tile instead
      const char *e = tune_env("SF_MT_WIDE_SMALL");
      return e ? atoi(e) : 176;   // the qkv projections of depths 6-7 at 16-32 evaluations per branch (1408 x 1536 x 1024: 88 tiles of 128x192)
    }();
    const long t128w = (long)((a.M + 127) / 128) * ((a.n_store + 127) / 128);
    if (wide && a.geom == 0 && t128w <= wide_small && a.n_store % 64 == 0 && a.K >= 512) return 7;
  }
  if (wide) return rule == 0 ? 2 : 6;   // in the U-Net step too the two-slot 128x192 tile wins on the qkv projections (28.6 vs 34.2 us)
  if (a.geom == 0) {
    // one workgroup per CU (the LDS ring fills it), each bound by its L2 -> LDS fill ~ (BM + BN) per K step: a launch costs
    // rounds x (BM + BN).  192-row tiles turn the 176-tile launches of the guidance batch (69 % of the CUs) into 235-240.
    const long nt = (a.n_store + 127) / 128;
    {
      // Launches that leave half of the CUs or more without a 128x128 tile (the deep levels at 16 evaluations per branch: 48-132
      // tiles) take 128x64 tiles, which fit twice on a CU.  Alone on the chip (tools/mt_d7_variants.sh, profiles/r3_f_mt_variants_*):
      // 1408 x 1024 x 3072 32.7 -> 23.5 us, 704 x 1024 x 3072 31.9 -> 22.8, 2816 x 512 x 1536 18.7 -> 14.0, 5632 x 256 x 768 12.0 -> 9.0,
      // 1408 x 1024 x 1280 17.0 -> 12.4, 1408 x 1536 x 1024 15.5 -> 14.0.  In the step: batch 32 without guidance 256.9 -> 263.2 steps/s
      // with the K >= 1536 launches alone, 266.0 -> 268.6 with the shorter reductions and the 132-tile launches added (alternating
      // runs, profiles/r3_f_ab_small_tiles*.txt).  The 176-tile launches of the guidance batch gain 4 % alone (34.1 -> 32.6 us) and
      // nothing in the step: the threshold stays below them.
      static const int thr = [] {   // tuning hook: largest 128x128 tile count that still switches to 128x64 (0 = never)
        const char *e = tune_env("SF_MT_SMALL_TILES");
        return e ? atoi(e) : 176;
      }();
      const long t128 = (long)((a.M + 127) / 128) * nt;
      static const int kmin = [] {   // tuning hook: shortest reduction that takes the rule
        const char *e = tune_env("SF_MT_SMALL_KMIN");
        return e ? atoi(e) : 256;   // 256 (was 512): + the InjectChannels GEMMs of depth 4; with SF_MT_WIDE_SMALL: configs[2] +1.4 %, batch 32 +3.8 % (profiles/r5_b_ab_tiles.txt)
      }();
      if (t128 <= thr && a.K >= kmin && a.n_store % 64 == 0 && rule != 0) return 7;
    }
    auto cost = [&](int bm) { return (((long)((a.M + bm - 1) / bm) * nt + 255) / 256) * (bm + 128); };
    const long c256 = cost(256), c192 = cost(192), c128 = cost(128);
    static const int v192 = [] { const char *e = tune_env("SF_MT_V192"); return e ? atoi(e) : 3; }();   // tuning hooks: 8 / 5 = the two-slot forms
    static const int v128 = [] { const char *e = tune_env("SF_MT_V128"); return e ? atoi(e) : 1; }();
    if (c192 < c256 && c192 <= c128) return v192;
    return c128 < c256 ? v128 : 0;
  }
  return t256 < 160 ? 1 : 0;                                   // few row bands: halve the tile so that more CUs get one
}

const char *conv_gemm_mt_name(const ConvGemmArgs &a) {
  static const char *n[11] = {"conv_gemm_mt<bf16,256x128>", "conv_gemm_mt<bf16,128x128>", "conv_gemm_mt<bf16,128x192>", "conv_gemm_mt<bf16,192x128>",
                             "conv_gemm_mt<bf16,256x64>", "conv_gemm_mt<bf16,128x128,2wg>", "conv_gemm_mt<bf16,128x192,2wg>", "conv_gemm_mt<bf16,128x64,2wg>",
                             "conv_gemm_mt<bf16,192x128,2wg>", "conv_gemm_mt<bf16,256x64,2wg>", "conv_gemm_mt<bf16,256x256>"};
  return n[conv_gemm_mt_variant(a)];
}

hipError_t launch_conv_gemm_mt(int dt, const ConvGemmArgs &a, hipStream_t s) {
  if (!conv_gemm_mt_ok(dt, a)) return hipErrorInvalidValue;
  if (dt == F32 && a.wx) {
    const int v = conv_gemm_mt_x3_variant(a);
    if (a.wx_mode == X3_BF16) return (a.cin2 || a.src_x3) ? hipErrorInvalidValue : launch_mt_x3<false, X3_BF16>(a, v, s);   // (gradient GEMMs have one source)
    if (a.src_x3) return a.cin2 ? hipErrorInvalidValue : launch_mt_x3<false, X3_F16, true>(a, v, s);
    return a.cin2 ? launch_mt_x3<true, X3_F16>(a, v, s) : launch_mt_x3<false, X3_F16>(a, v, s);
  }
  if (dt == F32) {
    static const int forced = [] { const char *e = tune_env("SF_MT_F32_VARIANT"); return e ? atoi(e) : -1; }();   // tuning hook: 1, 5, 7
    const long t128 = (long)((a.M + 127) / 128) * ((a.n_store + 127) / 128);
    const int v = forced >= 0 ? forced : ((a.n_store % 64 == 0 && t128 < 512) ? 7 : 5);
    return a.cin2 ? launch_mt_v<float, 0, true>(a, v, s) : launch_mt_v<float, 0, false>(a, v, s);
  }
  const int v = conv_gemm_mt_variant(a);
  if (dt == F16) return a.geom == 1 ? launch_mt_v<f16, 1, false>(a, v, s) : (a.cin2 ? launch_mt_v<f16, 0, true>(a, v, s) : launch_mt_v<f16, 0, false>(a, v, s));
  return a.geom == 1 ? launch_mt_v<bf16, 1, false>(a, v, s) : (a.cin2 ? launch_mt_v<bf16, 0, true>(a, v, s) : launch_mt_v<bf16, 0, false>(a, v, s));
}

}  // namespace sf
