// Channel-block split-K convolution ("cb") for the DEEP U-Net levels at small batch, and the two launches that consume it.
//
// What the wave-private 32x32 kernel (conv_gemm_wp.hip) pays at depths 5-7: a tile streams (32 + 32) * K * 2 B = 393 KB at
// K = 3072 through ONE CU, 192 such workgroups move 75 MB through the vector caches for 6.3 MB of weights (a 12-fold operand
// amplification at the chip's ~14 TB/s L2 -> CU ceiling), and its output feeds a launch that reads the whole tensor anyway
// (GroupNorm+SiLU, or LayerNorm+Modulation).  Here the reduction is split over WORKGROUPS along the input channels:
//
//   conv_cb        one workgroup = BM rows (32 * MT, flat over the clips) x 128 output columns x ONE 128-channel block of the
//                  input, all three taps of it (K slice = 3 * 128).  The activation panel ((BM + 2) rows x 128 channels) is read
//                  ONCE for the three taps, optionally GroupNorm+SiLU'd on the way into LDS (23-35 elements per thread: the
//                  prologue that cost +11 us on full-K tiles is 0.3 us here); the weight slice arrives in MFMA FRAGMENT ORDER
//                  (packed once per weight version: a wave's 16-byte-per-lane load is 1 KB contiguous), straight into registers,
//                  all 24 loads of a wave issued up front behind the panel loads -- they do not depend on the producer kernel.
//                  A workgroup streams 107-131 KB instead of 393 KB; d7 (M = 176): 192 workgroups x 115 KB = 22 MB.
//                  Output: fp32 partial slabs  slab[cb][m][n]  (no bias), 16-byte stores.
//   cb_reduce_gn   the launch that already followed conv1 (gn_silu) becomes the reducer: sum of the slabs + bias -> h (16-bit)
//                  and the GroupNorm chunk partials of h; >= 176 workgroups instead of 32.  The second convolution applies
//                  GroupNorm+SiLU from those partials in its panel prologue.
//   cb_reduce_ln   the launch that already followed conv2 (ln_modulate): sum of the slabs + bias + residual x -> y (never stored),
//                  LayerNorm over the row in fp32, Modulation -> m (16-bit).
// No tickets, no last-arriver, no extra kernel boundary: the reduction rides with launches the chain already had.
//
// Reference arithmetic: a-unet ResnetItem / ModulationItem (exp/model/diffusion.yaml:17-19).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int KC = 128;         // input channels per workgroup (one K block = 3 taps x 128)
constexpr int KC_LOG2 = 7;
constexpr int NSLOT = 4;        // clips a (BM + 2 <= 130)-row panel may touch: L >= 44 (host guarantees it)
constexpr unsigned OOB = 0x80000000u;

template <typename T> __device__ __forceinline__ Vec16<T> buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  Vec16<T> v;
  u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  v.v = __builtin_bit_cast(decltype(v.v), raw);
  return v;
}

// MT = 32-row MFMA tiles per workgroup.  256 threads: wave w owns output columns [32 w, 32 w + 32) of the 128-column block.
// PRO: the GroupNorm+SiLU panel prologue, a compile-time switch (as a run-time branch its loaded values meet "undefined" at the join
// and the compiler's copies put a wait in front of the weight loads).
// Grid: x = weight slice (column block, channel block), y = row tile (rows y >= mtiles: hosted prefetch).  Every division of the index
// arithmetic is a shift (S, channels per group, groups per block are powers of two) or a multiply-high by ceil(2^32 / L) (exact for
// x * L < 2^32): the weight loads are issued ~100 instructions into the kernel, not ~400.
// KB = 128-channel blocks per workgroup (1 or 2): two blocks halve the number of partial slabs (their traffic is what limits the chain at
// 8-16 clips per branch) for twice the weight stream per workgroup (48 fragments per wave).
template <typename T, int MT, bool PRO, int KB>
__global__ __launch_bounds__(256) void conv_cb_kernel(const ConvCbArgs a, const int mtiles, const unsigned bytes_src) {
  using frag = typename Frag16<T>::type;
  constexpr int KCW = KC * KB, PITCHW = KCW + 8, TPR = KCW / 8, RPP = 256 / TPR;   // channels per workgroup, LDS pitch, threads per row, rows per pass
  constexpr int BM = 32 * MT, PR = BM + 2, NV = (PR + RPP - 1) / RPP;
  __shared__ __attribute__((aligned(16))) T panel[PR * PITCHW];
  __shared__ float2 gstat[NSLOT][8];   // (mean, rstd) per (clip slot, group inside the channel block)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nws = (int)gridDim.x;
  if ((int)blockIdx.y >= mtiles) {   // hosted weight prefetch for the next GEMM of the chain (kernels.h, Prefetch)
    prefetch_slice(a.pf, ((int)blockIdx.y - mtiles) * nws + (int)blockIdx.x, 256);
    return;
  }
  // workgroups that share a weight slice (same x) are nws apart in dispatch order: with nws % 8 == 0 they land on the same XCD
  const int mt = (int)blockIdx.y, ws = (int)blockIdx.x;
  const int nt = ws >> a.log2S, cb = ws & ((1 << a.log2S) - 1);
  const int r0 = mt * BM, M = a.B * a.L;
  auto divL = [&](int x) { return (int)__umulhi((unsigned)x, a.magicL); };

  // ---- 1. activation panel: rows r0 - 1 .. r0 + BM of channel block cb (16 threads per row, 16 rows per pass) ----------------
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, bytes_src, 0x00020000);
  const int cv = tid % TPR, pr = tid / TPR;
  Vec16<T> pv[NV];
  const unsigned col_b = (unsigned)((cb * KCW + cv * 8) * sizeof(T)), row_b = (unsigned)(a.src_ld * sizeof(T));
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = pr + RPP * i, r = r0 - 1 + j;
    const bool ok = j < PR && r >= 0 && r < M;
    pv[i] = buf_ld16<T>(rS, ok ? (unsigned)r * row_b + col_b : OOB);
  }
  // ---- 2. GroupNorm operands: gamma / beta of this thread's channel octet, chunk sums of the clips the panel touches ----------
  //         (8 lanes per (clip, group) pair, 4 chunks each; every address is clamped so that the loads are branch-free and all in
  //         flight together, behind the panel loads and in front of the weight loads: the first wait must not cover the weights)
  const int gshift = KC_LOG2 + (KB == 2 ? 1 : 0) - a.log2cpg;   // log2(groups inside the workgroup's channel range)
  const int clip0 = divL(max(r0 - 1, 0));
  const int nclip = PRO ? (divL(min(r0 + BM, M - 1)) - clip0 + 1) : 0;
  const int npair = nclip << gshift;                           // <= NSLOT * 8 = 32: one pair per 8 lanes
  f32x4 ga[2], be[2];
  constexpr int NRD = 1;
  float2 cs[NRD][4];
  int lim[NRD] = {0};
  const int sub = tid & 7;
  if constexpr (PRO) {
    const int G = a.C >> a.log2cpg;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ga[q] = *reinterpret_cast<const f32x4 *>(a.gamma + cb * KCW + cv * 8 + 4 * q);
      be[q] = *reinterpret_cast<const f32x4 *>(a.beta + cb * KCW + cv * 8 + 4 * q);
    }
    // pro 1: chunk sums [clip][nch][G]; pro 2: tile sums [m tile][C / 32][segment] of the producing GEMM (kernels.h).  One address
    // select per load instead of two code paths: loaded values that meet at a join cost a wait in front of the weight loads.
    const bool tiles = a.pro == 2;
    const int ltpg = a.log2cpg - 5, ct = a.C >> 5;               // log2(32-column tiles per group), tiles per row
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int p = min((tid >> 3) + 32 * rd, npair - 1);
      const int slot = p >> gshift, gi = p & ((1 << gshift) - 1);
      const int clip = clip0 + slot;
      const int mt_lo = (clip * a.L) >> 5, nmt = ((clip * a.L + a.L - 1) >> 5) - mt_lo + 1;
      lim[rd] = tiles ? (nmt << max(ltpg, 0)) : a.nch;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int kc = min(sub + 8 * k, lim[rd] - 1);
        const int mtile = mt_lo + (kc >> max(ltpg, 0));
        const int tile = (((cb << gshift) + gi) << max(ltpg, 0)) + (kc & ((1 << max(ltpg, 0)) - 1));
        const unsigned off2 = (unsigned)(((mtile * ct + tile) * 2 + (clip - divL(mtile << 5))) * 2);
        const unsigned off1 = (unsigned)((((clip * a.nch + kc) * G) + (cb << gshift) + gi) * 2);
        cs[rd][k] = *reinterpret_cast<const float2 *>(a.stats + (tiles ? off2 : off1));
      }
    }
  }
  // ---- 3. the wave's weight slice, fragment order: 24 x 1 KB contiguous, independent of the producer kernel -----------------
  __builtin_amdgcn_sched_barrier(0);   // issue order = wait order (vmcnt counts in issue order): panel, GroupNorm operands, THEN weights
  frag wf[24 * KB];
#pragma unroll
  for (int blk = 0; blk < KB; ++blk) {
    const frag *wp = reinterpret_cast<const frag *>(a.wp) + ((size_t)((cb * KB + blk) * (a.N >> 5) + (nt * 4 + wave)) * 24) * 64;   // uniform
#pragma unroll
    for (int s = 0; s < 24; ++s) wf[blk * 24 + s] = wp[s * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  // (the compiler otherwise hoists the first use of a chunk sum in front of the weight loads, and its wait with it)
  if constexpr (PRO) {
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd)
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(cs[rd][k].x), "+v"(cs[rd][k].y));
  }
  // ---- 4. statistics of the touched (clip, group) pairs -> LDS ----------------------------------------------------------------
  if constexpr (PRO) {
    const float rn = 1.0f / ((float)a.L * (float)(1 << a.log2cpg));
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (sub + 8 * k < lim[rd]) {
          s1 += cs[rd][k].x;
          s2 += cs[rd][k].y;
        }
      s1 = sum8_dpp(s1);
      s2 = sum8_dpp(s2);
      const int p = (tid >> 3) + 32 * rd;
      if (sub == 0 && p < npair) {
        const float mean = s1 * rn;
        gstat[p >> gshift][p & ((1 << gshift) - 1)] = make_float2(mean, rsqrtf(fmaxf(fmaf(s2, rn, -mean * mean), 0.f) + a.eps));
      }
    }
    __syncthreads();
  }
  // ---- 5. panel -> LDS, GroupNorm + SiLU applied on the way -------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = pr + RPP * i, r = r0 - 1 + j;
    if (j < PR) {
      Vec16<T> o = pv[i];
      if (PRO && r >= 0 && r < M) {
        const float2 st = gstat[divL(r) - clip0][(cv * 8) >> a.log2cpg];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = st.y * ga[e >> 2][e & 3];
          const float sh = fmaf(-st.x, sc, be[e >> 2][e & 3]);
          o.set(e, silu_t<true>(fmaf(pv[i].get(e), sc, sh)));
        }
      }
      st16<T>(panel + j * PITCHW + cv * 8, o);
    }
  }
  __syncthreads();
  // ---- 6. D^T[n][m] += W[n][k] * act[m + tap - 1][k]: the weight fragment is the MFMA's A operand, so a lane ends up with four
  //         consecutive output COLUMNS of one row (16-byte slab stores) ---------------------------------------------------------
  const int fr = lane & 31, fh = lane >> 5;
  bool ok0[MT], ok2[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = r0 + t * 32 + fr, l = m - divL(m) * a.L;
    ok0[t] = l > 0;              // tap 0 reads position l - 1 of the same clip
    ok2[t] = l < a.L - 1;        // tap 2 reads position l + 1
  }
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const frag zf = __builtin_bit_cast(frag, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
  for (int blk = 0; blk < KB; ++blk)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          frag af = *reinterpret_cast<const frag *>(panel + (t * 32 + fr + tap) * PITCHW + blk * KC + ks * 16 + fh * 8);
          if (tap == 0) af = ok0[t] ? af : zf;
          if (tap == 2) af = ok2[t] ? af : zf;
          acc[t] = mfma32x16(wf[blk * 24 + tap * 8 + ks], af, acc[t]);
        }
  // ---- 7. partial slab ---------------------------------------------------------------------------------------------------------
  float *sl = a.slab + (size_t)cb * M * a.N + nt * 128 + wave * 32 + 4 * fh;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = r0 + t * 32 + fr;
    if (m < M) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        *reinterpret_cast<f32x4 *>(sl + (unsigned)(m * a.N + 8 * g)) = v;
      }
    }
  }
}

// The same workgroup on split fp16 operands (the fp32x engine: fp32 activations in HBM, fp32 partial slabs out).  The panel is fetched as
// fp32 (32 threads per row), GroupNorm+SiLU'd in fp32 and SPLIT ONCE per element on the way into two fp16 LDS images (hi, lo'): every
// fragment the three taps and the four waves read afterwards is a plain 16-byte row read, no vector instruction in the MFMA loop.  The
// weight slice arrives as (hi, lo') fragment pairs, 48 x 1 KB contiguous per wave; three MFMAs per product into two accumulators.
// One 128-channel block per workgroup (the two LDS images of a two-block panel would not fit the static 64 KB).
template <int MT, bool PRO>
__global__ __launch_bounds__(256) void conv_cb_x3_kernel(const ConvCbArgs a, const int mtiles, const unsigned bytes_src) {
  using frag = f16x8;
  constexpr int PITCHW = KC + 8, TPR = KC / 4, RPP = 256 / TPR;   // LDS pitch (fp16 elements), threads per panel row (16 B of fp32 each), rows per pass
  constexpr int BM = 32 * MT, PR = BM + 2, NV = (PR + RPP - 1) / RPP;
  __shared__ __attribute__((aligned(16))) f16 panelH[PR * PITCHW], panelL[PR * PITCHW];
  __shared__ float2 gstat[NSLOT][8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nws = (int)gridDim.x;
  if ((int)blockIdx.y >= mtiles) {
    prefetch_slice(a.pf, ((int)blockIdx.y - mtiles) * nws + (int)blockIdx.x, 256);
    return;
  }
  const int mt = (int)blockIdx.y, ws = (int)blockIdx.x;
  const int nt = ws >> a.log2S, cb = ws & ((1 << a.log2S) - 1);
  const int r0 = mt * BM, M = a.B * a.L;
  auto divL = [&](int x) { return (int)__umulhi((unsigned)x, a.magicL); };

  // ---- 1. activation panel (fp32): rows r0 - 1 .. r0 + BM of channel block cb ----------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, bytes_src, 0x00020000);
  const int cv = tid % TPR, pr = tid / TPR;
  f32x4 pv[NV];
  const unsigned col_b = (unsigned)((cb * KC + cv * 4) * 4), row_b = (unsigned)(a.src_ld * 4);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = pr + RPP * i, r = r0 - 1 + j;
    const bool ok = j < PR && r >= 0 && r < M;
    pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rS, ok ? (unsigned)r * row_b + col_b : OOB, 0, 0));
  }
  // ---- 2. GroupNorm operands (as in conv_cb_kernel; this thread owns a channel QUAD) ----------------------------------------------------
  const int gshift = KC_LOG2 - a.log2cpg;
  const int clip0 = divL(max(r0 - 1, 0));
  const int nclip = PRO ? (divL(min(r0 + BM, M - 1)) - clip0 + 1) : 0;
  const int npair = nclip << gshift;
  f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
  float2 cs[4];
  int lim = 0;
  const int sub = tid & 7;
  if constexpr (PRO) {
    const int G = a.C >> a.log2cpg;
    ga = *reinterpret_cast<const f32x4 *>(a.gamma + cb * KC + cv * 4);
    be = *reinterpret_cast<const f32x4 *>(a.beta + cb * KC + cv * 4);
    const bool tiles = a.pro == 2;
    const int ltpg = a.log2cpg - 5, ct = a.C >> 5;
    const int p = min(tid >> 3, npair - 1);
    const int slot = p >> gshift, gi = p & ((1 << gshift) - 1);
    const int clip = clip0 + slot;
    const int mt_lo = (clip * a.L) >> 5, nmt = ((clip * a.L + a.L - 1) >> 5) - mt_lo + 1;
    lim = tiles ? (nmt << max(ltpg, 0)) : a.nch;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kc = min(sub + 8 * k, lim - 1);
      const int mtile = mt_lo + (kc >> max(ltpg, 0));
      const int tile = (((cb << gshift) + gi) << max(ltpg, 0)) + (kc & ((1 << max(ltpg, 0)) - 1));
      const unsigned off2 = (unsigned)(((mtile * ct + tile) * 2 + (clip - divL(mtile << 5))) * 2);
      const unsigned off1 = (unsigned)((((clip * a.nch + kc) * G) + (cb << gshift) + gi) * 2);
      cs[k] = *reinterpret_cast<const float2 *>(a.stats + (tiles ? off2 : off1));
    }
  }
  // ---- 3. the wave's weight slice: 24 (hi, lo') fragment pairs, 48 KB contiguous per wave, independent of the producer kernel -----------
  __builtin_amdgcn_sched_barrier(0);
  frag wh[24], wl[24];
  {
    const frag *wp = reinterpret_cast<const frag *>(a.wp) + ((size_t)(cb * (a.N >> 5) + (nt * 4 + wave)) * 48) * 64;   // uniform
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      wh[s] = wp[(2 * s) * 64 + lane];
      wl[s] = wp[(2 * s + 1) * 64 + lane];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PRO) {
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(cs[k].x), "+v"(cs[k].y));
  }
  // ---- 4. statistics of the touched (clip, group) pairs -> LDS ---------------------------------------------------------------------------
  if constexpr (PRO) {
    const float rn = 1.0f / ((float)a.L * (float)(1 << a.log2cpg));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (sub + 8 * k < lim) {
        s1 += cs[k].x;
        s2 += cs[k].y;
      }
    s1 = sum8_dpp(s1);
    s2 = sum8_dpp(s2);
    const int p = tid >> 3;
    if (sub == 0 && p < npair) {
      const float mean = s1 * rn;
      gstat[p >> gshift][p & ((1 << gshift) - 1)] = make_float2(mean, rsqrtf(fmaxf(fmaf(s2, rn, -mean * mean), 0.f) + a.eps));
    }
    __syncthreads();
  }
  // ---- 5. panel -> LDS: GroupNorm + SiLU in fp32, then split into the two fp16 images ----------------------------------------------------
  typedef f16 f16x4_c __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = pr + RPP * i, r = r0 - 1 + j;
    if (j < PR) {
      f32x4 o = pv[i];
      if (PRO && r >= 0 && r < M) {
        const float2 st = gstat[divL(r) - clip0][(cv * 4) >> a.log2cpg];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = st.y * ga[e];
          const float sh = fmaf(-st.x, sc, be[e]);
          o[e] = silu_t<false>(fmaf(pv[i][e], sc, sh));
        }
      }
      f16x4_c h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f16 x, y;
        x3_split1<X3_F16>(o[e], x, y);
        h[e] = x;
        l[e] = y;
      }
      *reinterpret_cast<f16x4_c *>(panelH + j * PITCHW + cv * 4) = h;
      *reinterpret_cast<f16x4_c *>(panelL + j * PITCHW + cv * 4) = l;
    }
  }
  __syncthreads();
  // ---- 6. D^T[n][m] += W[n][k] * act[m + tap - 1][k] -------------------------------------------------------------------------------------
  const int fr = lane & 31, fh = lane >> 5;
  bool ok0[MT], ok2[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = r0 + t * 32 + fr, l = m - divL(m) * a.L;
    ok0[t] = l > 0;
    ok2[t] = l < a.L - 1;
  }
  f32x16 acc[MT], accL[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = accL[t][r] = 0.f;
  const frag zf = __builtin_bit_cast(frag, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
  for (int tap = 0; tap < 3; ++tap)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int o = (t * 32 + fr + tap) * PITCHW + ks * 16 + fh * 8;
        frag ah = *reinterpret_cast<const frag *>(panelH + o), al = *reinterpret_cast<const frag *>(panelL + o);
        if (tap == 0) {
          ah = ok0[t] ? ah : zf;
          al = ok0[t] ? al : zf;
        }
        if (tap == 2) {
          ah = ok2[t] ? ah : zf;
          al = ok2[t] ? al : zf;
        }
        x3_mfma<X3_F16>(wh[tap * 8 + ks], wl[tap * 8 + ks], ah, al, acc[t], accL[t]);
      }
  // ---- 7. partial slab --------------------------------------------------------------------------------------------------------------------
  float *sl = a.slab + (size_t)cb * M * a.N + nt * 128 + wave * 32 + 4 * fh;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = r0 + t * 32 + fr;
    if (m < M) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(accL[t][4 * g + e], X3P<X3_F16>::INV, acc[t][4 * g + e]);
        *reinterpret_cast<f32x4 *>(sl + (unsigned)(m * a.N + 8 * g)) = v;
      }
    }
  }
}

// the same weights as (hi, lo') fragment pairs for conv_cb_x3_kernel: [cb][n / 32][tap][ks][part][lane][8]
__global__ void pack_conv_cb_x3_kernel(const float *__restrict__ w, int N, int C, f16 *__restrict__ out) {
  const size_t total = (size_t)N * C * 3;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
    size_t t = e >> 9;
    const int ks = (int)(t & 7);
    t >>= 3;
    const int tap = (int)(t % 3);
    t /= 3;
    const int nt32 = (int)(t % (N / 32)), cb = (int)(t / (N / 32));
    const int n = nt32 * 32 + (lane & 31), c = cb * KC + ks * 16 + (lane >> 5) * 8 + q;
    const size_t pair = (e >> 9) * 1024;   // 512 hi then 512 lo' per (cb, n / 32, tap, ks)
    x3_split1<X3_F16>(w[((size_t)n * C + c) * 3 + tap], out[pair + (e & 511)], out[pair + 512 + (e & 511)]);
  }
}

// Conv1d weight (N, C, 3) fp32 -> fragment order [cb][n / 32][tap][ks][lane][8]:
//   lane holds W[n = 32 * (n / 32) + lane % 32][tap][c = 128 cb + 16 ks + 8 (lane / 32) + 0..7]
template <typename T>
__global__ void pack_conv_cb_kernel(const float *__restrict__ w, int N, int C, T *__restrict__ out) {
  const size_t total = (size_t)N * C * 3;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
    size_t t = e >> 9;
    const int ks = (int)(t & 7);
    t >>= 3;
    const int tap = (int)(t % 3);
    t /= 3;
    const int nt32 = (int)(t % (N / 32)), cb = (int)(t / (N / 32));
    const int n = nt32 * 32 + (lane & 31), c = cb * KC + ks * 16 + (lane >> 5) * 8 + q;
    out[e] = from_f<T>(w[((size_t)n * C + c) * 3 + tap]);
  }
}

// slabs -> h = sum + bias (16-bit) and GroupNorm chunk sums (sum, sum of squares per (clip, chunk, group)) of the STORED h.  Workgroup = (clip, chunk of 8 P rows, 128-column block);
// thread = 4 consecutive columns of one row per pass.
template <typename T, int S, int P>
__global__ __launch_bounds__(256) void cb_reduce_gn_kernel(const float *__restrict__ slab, const int M, const int N, const int L,
                                                           const float *__restrict__ bias, T *__restrict__ out, const int out_ld, const int G,
                                                           float *__restrict__ stats, const int nch, const int nreal, const Prefetch pf) {
  __shared__ float red[4][8][2];
  if ((int)blockIdx.x >= nreal) {
    prefetch_slice(pf, (int)blockIdx.x - nreal, 256);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = N / 128;
  const int cblk = (int)blockIdx.x % ncb, bc = (int)blockIdx.x / ncb;
  const int b = bc / nch, ch = bc - b * nch;
  const int cq = tid & 31, rr = tid >> 5;
  const int c = cblk * 128 + cq * 4;
  const int l0 = ch * 8 * P, rows = min(8 * P, L - l0);
  f32x4 v[P][S];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int lr = p * 8 + rr;
    const size_t m = (size_t)b * L + l0 + min(lr, rows - 1);
#pragma unroll
    for (int s = 0; s < S; ++s) v[p][s] = *reinterpret_cast<const f32x4 *>(slab + ((size_t)s * M + m) * N + c);
  }
  const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + c);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int lr = p * 8 + rr;
    f32x4 x = v[p][0];
#pragma unroll
    for (int s = 1; s < S; ++s) x += v[p][s];   // fixed order: deterministic
    x += bi;
    if (lr < rows) {
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = from_f<T>(x[e]);
        const float xo = to_f(o[e]);
        s1 += xo;
        s2 = fmaf(xo, xo, s2);
      }
      __builtin_memcpy(__builtin_assume_aligned(out + ((size_t)b * L + l0 + lr) * out_ld + c, 4 * sizeof(T)), o, 4 * sizeof(T));
    }
  }
  // group totals: lanes of one group are `span` consecutive channel quads; lane ^ 32 is the same quad one row further
  const int cpg = N / G, span = cpg / 4;   // 4 .. 32
  for (int off = 1; off < span; off <<= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  const int gpb = 128 / cpg;
  if (lane < 32 && (cq % span) == 0) {
    red[wave][cq / span][0] = s1;
    red[wave][cq / span][1] = s2;
  }
  __syncthreads();
  if (tid < gpb) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a1 += red[w][tid][0];
      a2 += red[w][tid][1];
    }
    float *o = stats + (((size_t)b * nch + ch) * G + cblk * gpb + tid) * 2;   // (sum, sum of squares): the consumer just adds the chunks
    o[0] = a1;
    o[1] = a2;
  }
}

// slabs + bias + residual -> y (fp32, never stored);  m = LayerNorm_C(y; eps) * (1 + ss[b][c]) + ss[b][C + c]  -> out (16-bit).
// TPR = C / 4 threads per row (32 .. 256), 256 / TPR rows per workgroup.
template <typename T, int S, int TPR>
__global__ __launch_bounds__(256) void cb_reduce_ln_kernel(const float *__restrict__ slab, const int M, const int C, const int L,
                                                           const float *__restrict__ bias, const T *__restrict__ res, const int res_ld,
                                                           const float *__restrict__ ss, const int ss_ld, const float eps, T *__restrict__ out,
                                                           const int out_ld, const int nreal, const Prefetch pf) {
  constexpr int RPB = 256 / TPR, WPR = TPR >= 64 ? TPR / 64 : 1;   // rows per workgroup, waves per row
  __shared__ float red[2][4];
  if ((int)blockIdx.x >= nreal) {
    prefetch_slice(pf, (int)blockIdx.x - nreal, 256);
    return;
  }
  const int tid = threadIdx.x, wave = tid >> 6;
  const int row = (int)blockIdx.x * RPB + tid / TPR, sub = tid % TPR;
  const bool active = row < M;
  const size_t m = active ? row : 0;
  const int c = sub * 4, b = (int)(m / L);
  f32x4 v[S];
#pragma unroll
  for (int s = 0; s < S; ++s) v[s] = *reinterpret_cast<const f32x4 *>(slab + ((size_t)s * M + m) * C + c);
  T rv[4];
  __builtin_memcpy(rv, __builtin_assume_aligned(res + m * res_ld + c, 4 * sizeof(T)), 4 * sizeof(T));
  const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias + c);
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (ss) {
    sc = *reinterpret_cast<const f32x4 *>(ss + (size_t)b * ss_ld + c);
    sh = *reinterpret_cast<const f32x4 *>(ss + (size_t)b * ss_ld + C + c);
  }
  f32x4 y = v[0];
#pragma unroll
  for (int s = 1; s < S; ++s) y += v[s];
  y += bi;
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] += to_f(rv[e]);
  auto row_sum = [&](float x, int slot) {
    constexpr int W = TPR < 64 ? TPR : 64;
#pragma unroll
    for (int o = W >> 1; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if constexpr (WPR > 1) {
      if ((tid & 63) == 0) red[slot][wave] = x;
      __syncthreads();
      const int w0 = (wave / WPR) * WPR;
      x = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) x += red[slot][w0 + w];
    }
    return x;
  };
  const float mean = row_sum((y[0] + y[1]) + (y[2] + y[3]), 0) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float d = y[e] - mean;
    q = fmaf(d, d, q);
  }
  const float rstd = rsqrtf(row_sum(q, 1) / (float)C + eps);
  if (!active) return;
  T o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float z = (y[e] - mean) * rstd;
    if (ss) z = fmaf(z, 1.0f + sc[e], sh[e]);
    o[e] = from_f<T>(z);
  }
  __builtin_memcpy(__builtin_assume_aligned(out + (size_t)row * out_ld + c, 4 * sizeof(T)), o, 4 * sizeof(T));
}

int cb_min_wgs() {
  static const int v = [] {   // tuning hook: fewest workgroups a launch should have when choosing the rows per workgroup
    const char *e = tune_env("SF_CB_MIN_WGS");
    return e && atoi(e) > 0 ? atoi(e) : 160;
  }();
  return v;
}

}  // namespace

size_t conv_cb_weight_elems(int N, int C) { return (size_t)N * C * 3; }

bool conv_cb_shape_ok(int dt, int B, int L, int C, int N, int G) {
  if (dt == F32) return false;   // (F32X: the split-operand form)
  if (C < KC || (C % KC) || (N % 128) || C / KC > 8 || ((C / KC) & (C / KC - 1))) return false;
  if (L < 44 || (int64_t)B * L * (int64_t)(C > N ? C : N) * 4 >= 0x7FFFFFF0ll) return false;   // a 130-row panel touches <= NSLOT clips
  if ((uint64_t)B * L * (uint64_t)L >= (1ull << 32)) return false;   // row / L by multiply-high (magicL) is exact only while row * L < 2^32
  if (G > 0) {
    if (C % G) return false;
    const int cpg = C / G;
    if (cpg < 16 || cpg > KC || (cpg & (cpg - 1))) return false;
  }
  return true;
}

// pro 2 (GroupNorm sums per 32 x 32 tile of the producing GEMM): a group must be whole tiles, a (clip, group) at most 32 of them
bool conv_cb_tile_stats_ok(int L, int C, int G) {
  if (G < 1 || C % G) return false;
  const int cpg = C / G;
  if (cpg < 32 || (cpg & (cpg - 1)) || L < 32) return false;
  return ((L + 30) / 32 + 1) * (cpg / 32) <= 32;
}

// Measured alone on the chip (tools/cb_bench.py, profiles/r4_a_cb_bench.txt; HBM-cold weights, us per launch, MT = 1 / 2 / 3 / 4):
//   d7 (176 rows, C 1024): 6.7 / 6.4 / 8.2 / 8.6     d6 (352 rows): 8.9 / 8.5 / 9.2 / 9.3     d5 (704 rows, C 512): 6.0 / 6.1 / 7.8 / 8.2
//   d4 (1408 rows, C 256): 4.0 / 5.1 / 6.8 / 7.1   (88 workgroups at MT = 2)
// 96- and 128-row panels hold 1 wave per SIMD (184-258 registers) and serialise a longer prologue: two tiles where that still
// gives >= 160 workgroups, else one.
int conv_cb_mt(int M, int N, int C) {
  static const int forced = [] {
    const char *e = tune_env("SF_CB_MT");
    return e ? atoi(e) : 0;
  }();
  if (forced >= 1 && forced <= 4) return forced;
  const int nws = (N / 128) * (C / KC);
  return ((M + 63) / 64) * nws >= cb_min_wgs() ? 2 : 1;
}

hipError_t launch_pack_conv_cb(int dt, const float *w, int N, int C, void *out, hipStream_t s) {
  if (dt == F32 || (C % KC) || (N % 32)) return hipErrorInvalidValue;
  const size_t total = (size_t)N * C * 3;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  if (dt == F32X) hipLaunchKernelGGL(pack_conv_cb_x3_kernel, dim3(blocks), dim3(256), 0, s, w, N, C, static_cast<f16 *>(out));
  else if (dt == BF16) hipLaunchKernelGGL((pack_conv_cb_kernel<bf16>), dim3(blocks), dim3(256), 0, s, w, N, C, static_cast<bf16 *>(out));
  else hipLaunchKernelGGL((pack_conv_cb_kernel<f16>), dim3(blocks), dim3(256), 0, s, w, N, C, static_cast<f16 *>(out));
  return hipGetLastError();
}

hipError_t launch_conv_cb(int dt, const ConvCbArgs &a0, hipStream_t s) {
  if (!conv_cb_shape_ok(dt, a0.B, a0.L, a0.C, a0.N, a0.pro ? a0.G : 0)) return hipErrorInvalidValue;
  if (a0.pro == 1 && a0.nch > 32) return hipErrorInvalidValue;
  if (a0.pro == 2 && !conv_cb_tile_stats_ok(a0.L, a0.C, a0.G)) return hipErrorInvalidValue;
  ConvCbArgs a = a0;
  if (dt == F32X && a.kb == 2) return hipErrorInvalidValue;   // one channel block per workgroup in the split-operand form
  const int kb = a.kb == 2 ? 2 : 1;
  const int M = a.B * a.L, S = a.C / KC / kb;   // partial slabs = workgroups along the input channels
  if (kb == 2 && ((a.C / KC) % 2 || (a.pro && a.C / a.G < 32))) return hipErrorInvalidValue;   // <= 8 groups inside a 256-channel range
  a.log2S = 0;
  while ((1 << a.log2S) < S) ++a.log2S;
  const int cpg = a.pro ? a.C / a.G : KC;
  a.log2cpg = 0;
  while ((1 << a.log2cpg) < cpg) ++a.log2cpg;
  if ((1 << a.log2S) != S || (1 << a.log2cpg) != cpg) return hipErrorInvalidValue;
  a.magicL = (unsigned)((0x100000000ull + (unsigned)a.L - 1) / (unsigned)a.L);   // x / L == mulhi(x, magicL) while x * L < 2^32
  int mt = kb == 2 ? std::min(2, conv_cb_mt(M, a.N, a.C / 2)) : conv_cb_mt(M, a.N, a.C);
  if (dt == F32X) mt = std::min(mt, 2);
  const int mtiles = (M + 32 * mt - 1) / (32 * mt), nws = (a.N / 128) * S;
  const int pf_rows = (a.pf.ptr && a.pf.bytes >= 16 && a.pf.wgs > 0) ? (a.pf.wgs + nws - 1) / nws : 0;
  a.pf.wgs = pf_rows * nws;
  const dim3 grid(nws, mtiles + pf_rows);
  const unsigned bytes_src = (unsigned)((size_t)M * a.src_ld * dsize(dt));
  if (dt == F32X) {
    if (a.pro) {
      if (mt == 1) hipLaunchKernelGGL((conv_cb_x3_kernel<1, true>), grid, dim3(256), 0, s, a, mtiles, bytes_src);
      else hipLaunchKernelGGL((conv_cb_x3_kernel<2, true>), grid, dim3(256), 0, s, a, mtiles, bytes_src);
    } else {
      if (mt == 1) hipLaunchKernelGGL((conv_cb_x3_kernel<1, false>), grid, dim3(256), 0, s, a, mtiles, bytes_src);
      else hipLaunchKernelGGL((conv_cb_x3_kernel<2, false>), grid, dim3(256), 0, s, a, mtiles, bytes_src);
    }
    return hipGetLastError();
  }
#define SF_CB2(T, MT, KB_)                                                                                                \
  do {                                                                                                                    \
    if (a.pro) hipLaunchKernelGGL((conv_cb_kernel<T, MT, true, KB_>), grid, dim3(256), 0, s, a, mtiles, bytes_src);       \
    else hipLaunchKernelGGL((conv_cb_kernel<T, MT, false, KB_>), grid, dim3(256), 0, s, a, mtiles, bytes_src);            \
  } while (0)
#define SF_CB(T, MT)                 \
  do {                               \
    if (kb == 2) SF_CB2(T, MT, 2);   \
    else SF_CB2(T, MT, 1);           \
  } while (0)
#define SF_CB_T(T)            \
  switch (mt) {               \
    case 1: SF_CB(T, 1); break; \
    case 2: SF_CB(T, 2); break; \
    case 3: if (kb == 2) { SF_CB2(T, 2, 2); } else { SF_CB2(T, 3, 1); } break; \
    default: if (kb == 2) { SF_CB2(T, 2, 2); } else { SF_CB2(T, 4, 1); } break; \
  }
  if (dt == BF16) { SF_CB_T(bf16) } else { SF_CB_T(f16) }
#undef SF_CB_T
#undef SF_CB
#undef SF_CB2
  return hipGetLastError();
}

// rows per statistics chunk of cb_reduce_gn: 8 P with the fewest passes that keep the chunk count within the 32 the consumers merge
CbGnPlan cb_gn_plan(int L) {
  CbGnPlan p;
  int P = 1;
  while (P < 4 && (L + 8 * P - 1) / (8 * P) > 32) P *= 2;
  p.chunk_rows = 8 * P;
  p.nch = (L + p.chunk_rows - 1) / p.chunk_rows;
  return p;
}

hipError_t launch_cb_reduce_gn(int dt, const float *slab, int S, int B, int L, int N, const float *bias, void *out, int out_ld, int G, float *stats,
                               const CbGnPlan &gp, hipStream_t s, Prefetch pf) {
  if ((N % 128) || N % G || (N / G) < 16 || (N / G) > 128 || (128 % (N / G)) || gp.nch > 32 || (out_ld % 4)) return hipErrorInvalidValue;
  const int P = gp.chunk_rows / 8, M = B * L;
  const int nreal = B * gp.nch * (N / 128);
  const dim3 grid(nreal + (pf.ptr && pf.bytes >= 16 ? pf.wgs : 0));
#define SF_RG(T, S_, P_)                                                                                                                          \
  hipLaunchKernelGGL((cb_reduce_gn_kernel<T, S_, P_>), grid, dim3(256), 0, s, slab, M, N, L, bias, static_cast<T *>(out), out_ld, G, stats, gp.nch, \
                     nreal, pf)
#define SF_RG_P(T, S_)            \
  switch (P) {                    \
    case 1: SF_RG(T, S_, 1); break; \
    case 2: SF_RG(T, S_, 2); break; \
    case 4: SF_RG(T, S_, 4); break; \
    default: return hipErrorInvalidValue; \
  }
#define SF_RG_S(T)                \
  switch (S) {                    \
    case 1: SF_RG_P(T, 1) break;  \
    case 2: SF_RG_P(T, 2) break;  \
    case 4: SF_RG_P(T, 4) break;  \
    case 8: SF_RG_P(T, 8) break;  \
    default: return hipErrorInvalidValue; \
  }
  if (dt == BF16) { SF_RG_S(bf16) } else if (dt == F16) { SF_RG_S(f16) } else { SF_RG_S(float) }   // float: the fp32x engine's chain
#undef SF_RG_S
#undef SF_RG_P
#undef SF_RG
  return hipGetLastError();
}

hipError_t launch_cb_reduce_ln(int dt, const float *slab, int S, int B, int L, int C, const float *bias, const void *res, int res_ld, const float *ss,
                               int ss_ld, float eps, void *out, int out_ld, hipStream_t s, Prefetch pf) {
  const int tpr = C / 4;
  if ((C % 128) || (tpr != 32 && tpr != 64 && tpr != 128 && tpr != 256) || (res_ld % 4) || (out_ld % 4)) return hipErrorInvalidValue;
  if (ss && ((ss_ld % 4) || (reinterpret_cast<uintptr_t>(ss) % 16))) return hipErrorInvalidValue;
  const int M = B * L, rpb = 256 / tpr;
  const int nreal = (M + rpb - 1) / rpb;
  const dim3 grid(nreal + (pf.ptr && pf.bytes >= 16 ? pf.wgs : 0));
#define SF_RL(T, S_, TPR_)                                                                                                                     \
  hipLaunchKernelGGL((cb_reduce_ln_kernel<T, S_, TPR_>), grid, dim3(256), 0, s, slab, M, C, L, bias, static_cast<const T *>(res), res_ld, ss, ss_ld, \
                     eps, static_cast<T *>(out), out_ld, nreal, pf)
#define SF_RL_T(T, S_)               \
  switch (tpr) {                     \
    case 32: SF_RL(T, S_, 32); break;  \
    case 64: SF_RL(T, S_, 64); break;  \
    case 128: SF_RL(T, S_, 128); break; \
    default: SF_RL(T, S_, 256); break; \
  }
#define SF_RL_S(T)                 \
  switch (S) {                     \
    case 1: SF_RL_T(T, 1) break;   \
    case 2: SF_RL_T(T, 2) break;   \
    case 4: SF_RL_T(T, 4) break;   \
    case 8: SF_RL_T(T, 8) break;   \
    default: return hipErrorInvalidValue; \
  }
  if (dt == BF16) { SF_RL_S(bf16) } else if (dt == F16) { SF_RL_S(f16) } else { SF_RL_S(float) }
#undef SF_RL_S
#undef SF_RL_T
#undef SF_RL
  return hipGetLastError();
}

}  // namespace sf
