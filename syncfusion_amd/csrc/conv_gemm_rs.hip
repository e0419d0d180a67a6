// Register-staged ("rs") implicit GEMM for the SHORT activations of the deep U-Net levels at small batch: 32x32 tiles, 16-bit types,
// K <= 2048.  Same tiling, wave-split K and epilogue as the wave-private kernel (conv_gemm_wp.hip), but NO staging and no pipeline:
//
//   * the weights come from a second, FRAGMENT-ORDERED copy of the matrix (ConvGemmArgs::wfr = [N / 32][K / 16][64 lanes][8]): the
//     16 bytes a lane needs for one 32x32x16 MFMA are where its load finds them, a wave's load is 1 KB contiguous, nothing passes
//     through LDS;
//   * the activations are loaded in fragment order too, straight from the rows (lane = row, 16 bytes = 8 consecutive k; the 128-byte
//     line of a row serves four consecutive fragments from the vector L1);
//   * every wave issues ALL loads of its K quarter up front (<= 32 + 32 fragments = 256 registers), weights interleaved with
//     activations, and multiplies as they land (counted vmcnt).
//
// The wave-private kernel keeps two 16 KB register sets per wave in flight and so pays one memory round trip per 128 of K per wave
// (K = 1280: ten dependent round trips, 8.3 us); here the whole 100-160 KB operand stream of a tile is in flight at once and the launch
// is bound by the CU's fill rate (~70 GB/s -> ~2 us).  Used for the 1x1 InjectChannels convolution over cat[x, ctx], the attention
// projections, and the patchify / up convolutions that fit (a-unet InjectChannelsItem / AttentionItem / Downsample).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr unsigned OOB = 0x80000000u;
constexpr int LDR = 36;   // row pitch of a partial tile in LDS (floats)

// T1: one tap (1x1 / Linear), a compile-time switch so that the two address schemes do not meet at a join
// X3 (T = float, the fp32x engine): fp32 activation fragments (two 16-byte loads per lane and 16 k), weights from the fragment-ordered
// split image ConvGemmArgs::wfrx = [N / 32][K / 16][hi | lo'][64 lanes][8] (2 KB contiguous per wave and fragment); every fragment is split
// in registers when it is multiplied (common.h, X3P<X3_F16>), three MFMAs per product, two accumulators.  16 registers per fragment in
// flight: K <= 1280.
template <typename T, bool CAT, int NFD, bool T1, bool X3 = false>
__global__ __launch_bounds__(256) void conv_gemm_rs_kernel(const ConvGemmArgs a, const int mtiles, const int ntiles, const int swz, const unsigned bytesA,
                                                           const unsigned bytesA2, const unsigned bytesW) {
  using frag = typename std::conditional<X3, f16x8, typename Frag16<T>::type>::type;
  constexpr int ES = X3 ? 4 : 2;
  static_assert(!X3 || sizeof(T) == 4, "split mode: fp32 activations");
  __shared__ __attribute__((aligned(16))) float red[4 * 32 * LDR + 4 * 32];   // four partial tiles | (mean, rstd) per row | (sum, sumsq) per row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= mtiles * ntiles) {   // hosted weight prefetch for the next GEMM of the chain (kernels.h, Prefetch)
    prefetch_slice(a.pf, (int)blockIdx.x - mtiles * ntiles, 256);
    return;
  }
  int bid = blockIdx.x, mt, nt;
  if (swz) {   // column tiles of one XCD share their weights' L2
    const int xcd = bid & 7, j = bid >> 3;
    nt = xcd + 8 * (j / mtiles);
    mt = j % mtiles;
  } else {
    nt = bid / mtiles;
    mt = bid % mtiles;
  }
  const int m0 = mt * 32, n0 = nt * 32;
  const int fr = lane & 31, fh = lane >> 5;

  const bool has_res = a.res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
  const bool ln_epi = a.ln_colsum != nullptr;
  const int ml = tid >> 3, nq = tid & 7;                 // epilogue: one thread = 4 consecutive columns of one row
  const int em = m0 + ml, enb = n0 + nq * 4;

  // ---- the wave's K quarter: all fragments in flight ----------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(CAT ? a.src2 : a.src), 0, CAT ? bytesA2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(X3 ? a.wfrx : a.wfr), 0, bytesW, 0x00020000);
  const int F = a.K >> 4, nf = F >> 2, f0 = wave * nf;          // K % 64 == 0: every wave takes F / 4 fragments of 16 k
  f32x16 acc, accL;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = accL[r] = 0.f;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  f32x4 eb, es_, ea, ec;
  u32x2 er, lp[4];
  u32x4 er4 = {0u, 0u, 0u, 0u};   // split mode: the residual quad in fp32
  {
    const int m = m0 + fr;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int b = mm / a.Lout, l = mm - b * a.Lout;
    const int rb = b * a.Lsrc, p0 = l * a.stride - a.pad;
    const int pmax = (a.Lsrc << a.up_shift) - 1;
    const unsigned vmask = vm ? 0u : OOB, lane_b = (unsigned)(fh * 8 * ES);
    const unsigned wbase = X3 ? (unsigned)((((size_t)nt * F + f0) * 128 + lane) * 16) : (unsigned)((((size_t)nt * F + f0) * 64 + lane) * 16);
    constexpr unsigned WSTEP = X3 ? 2048u : 1024u;   // bytes of one weight fragment (pair)
    const int k_taps = a.taps * a.cin;
    // (tap, channel) of the wave's first fragment, then streamed: all of it wave-uniform (scalar registers)
    int k = f0 * 16;
    int tap = (a.taps > 1 && k < k_taps) ? k / a.cin : 0;
    int c = k - tap * a.cin;
    auto row_off = [&](int t) {
      const int p = p0 + t;
      const unsigned bad = ((unsigned)p > (unsigned)pmax) ? OOB : 0u;
      return ((unsigned)(((rb + (max(p, 0) >> a.up_shift)) * a.src_ld) * ES) + lane_b) | bad | vmask;
    };
    unsigned cur = row_off(tap);
    const unsigned cur2 = CAT ? (((unsigned)(mm * a.src2_ld * ES) + lane_b) | vmask) : OOB;
    frag af[X3 ? 1 : NFD], wf[NFD];
    frag wl[X3 ? NFD : 1];            // split mode: the lo' halves of the weight fragments ...
    f32x4 ax[X3 ? NFD : 1][2];        // ... and the activation fragments as fp32 (8 consecutive k per lane)
    if constexpr (T1) {
      // 1x1 / Linear (+ concatenated second source): no tap bookkeeping, the k offset of fragment i is an instruction immediate
      const int kk = f0 * 16;
#pragma unroll
      for (int i = 0; i < NFD; ++i) {
        const unsigned dead = i < nf ? 0u : OOB;   // fragments past the wave's range read zeros (no memory traffic)
        const bool sec = CAT && (kk + 16 * i) >= k_taps;   // wave-uniform: scalar selects, one load instruction
        const unsigned off = sec ? cur2 + (unsigned)((kk - k_taps) * ES) : cur + (unsigned)(kk * ES);
        const __amdgpu_buffer_rsrc_t rs = sec ? rA2 : rA;
        if constexpr (X3) {
          ax[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (off + (unsigned)(i * 16 * ES)) | dead, 0, 0));
          ax[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (off + (unsigned)(i * 16 * ES) + 16u) | dead, 0, 0));
          wl[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rF, (wbase + (unsigned)(i * WSTEP) + 1024u) | dead, 0, 0));
        } else {
          af[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rs, (off + (unsigned)(i * 16 * ES)) | dead, 0, 0));
        }
        wf[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rF, (wbase + (unsigned)(i * WSTEP)) | dead, 0, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NFD; ++i) {
        const unsigned dead = i < nf ? 0u : OOB;
        if constexpr (X3) {
          if (CAT && k >= k_taps) {
            const unsigned aoff = (cur2 + (unsigned)((k - k_taps) * ES)) | dead;
            ax[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA2, aoff, 0, 0));
            ax[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA2, aoff + 16u, 0, 0));
          } else {
            const unsigned aoff = (cur + (unsigned)(c * ES)) | dead;
            ax[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff, 0, 0));
            ax[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, aoff + 16u, 0, 0));
          }
          wl[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rF, (wbase + (unsigned)(i * WSTEP) + 1024u) | dead, 0, 0));
        } else {
          if (CAT && k >= k_taps)
            af[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rA2, (cur2 + (unsigned)((k - k_taps) * ES)) | dead, 0, 0));
          else af[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rA, (cur + (unsigned)(c * ES)) | dead, 0, 0));
        }
        wf[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rF, (wbase + (unsigned)(i * WSTEP)) | dead, 0, 0));
        k += 16;
        c += 16;
        if (c >= a.cin && k < k_taps) {
          c = 0;
          ++tap;
          cur = row_off(tap);
        }
      }
    }
    // ---- epilogue operands: issued BEHIND the operand stream (they are needed last), branch-free -- an absent operand is a buffer
    //      resource of zero records, whose loads return zeros without touching memory (a `ptr ? load : 0` compiles to a branch with
    //      the load and an s_waitcnt vmcnt(0) inside it: five serial round trips in front of everything else)
    {
      auto rsrc = [](const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? 0x7FFFFFF0u : 0u, 0x00020000); };
      const int mc = min(em, a.M - 1), nc = min(enb, a.N - 4);
      const int bb = (has_bs || has_ba) ? mc / a.Lout : 0;
      eb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.bias), (unsigned)(nc * 4), 0, 0));
      es_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.bscale), (unsigned)((bb * a.bscale_ld + nc) * 4), 0, 0));
      ea = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.badd), (unsigned)((bb * a.badd_ld + nc) * 4), 0, 0));
      ec = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.ln_colsum), (unsigned)(nc * 4), 0, 0));
      if constexpr (X3) er4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.res), (unsigned)((mc * a.res_ld + nc) * ES), 0, 0);
      else er = __builtin_amdgcn_raw_buffer_load_b64(rsrc(a.res), (unsigned)((mc * a.res_ld + nc) * ES), 0, 0);
      const __amdgpu_buffer_rsrc_t rl = rsrc(a.ln_part);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pidx = nq + 8 * j;
        lp[j] = __builtin_amdgcn_raw_buffer_load_b64(rl, pidx < a.ln_nt ? (unsigned)((mc * a.ln_nt + pidx) * 8) : OOB, 0, 0);
      }
    }
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < NFD; ++i) {
        f16x8 ah, al;
        x3_split<X3_F16>(ax[i][0], ax[i][1], ah, al);
        x3_mfma<X3_F16>(ah, al, wf[i], wl[i], acc, accL);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaf(accL[r], X3P<X3_F16>::INV, acc[r]);
    } else {
#pragma unroll
      for (int i = 0; i < NFD; ++i) acc = mfma32x16(af[i], wf[i], acc);
    }
  }
  float bi[4], rv[4], sv[4], av[4], cu[4], ln_mp[4], ln_qp[4];
  {
    T rt[4];
    if constexpr (X3) __builtin_memcpy(rt, &er4, 16);
    else __builtin_memcpy(rt, &er, 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bi[e] = eb[e];
      sv[e] = has_bs ? es_[e] : 1.f;
      av[e] = ea[e];
      cu[e] = ec[e];
      rv[e] = has_res ? to_f(rt[e]) : 0.f;
      ln_mp[e] = __builtin_bit_cast(float, lp[e][0]);
      ln_qp[e] = __builtin_bit_cast(float, lp[e][1]);
    }
  }

  // ---- the four partial tiles meet here (fixed order: deterministic) ------------------------------------------------------------
  float *rowstat = red + 4 * 32 * LDR, *gsum = rowstat + 64;
  if (ln_epi) {
    const float mean = sum8_dpp((ln_mp[0] + ln_mp[1]) + (ln_mp[2] + ln_mp[3])) / (float)a.ln_nt;   // every partial covers 32 channels
    float dq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (nq + 8 * j < a.ln_nt) {
        const float d = ln_mp[j] - mean;
        dq += fmaf(32.f * d, d, ln_qp[j]);
      }
    const float m2 = sum8_dpp(dq);
    if (nq == 0) {
      rowstat[2 * ml] = mean;
      rowstat[2 * ml + 1] = rsqrtf(m2 / (float)a.cin + a.ln_eps);
    }
  }
  float *myred = red + wave * 32 * LDR;
#pragma unroll
  for (int r = 0; r < 16; ++r) myred[((r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + fr] = acc[r];
  __syncthreads();

  const bool live = em < a.M && enb < a.n_store;
  f32x4 v = *reinterpret_cast<const f32x4 *>(red + ml * LDR + nq * 4);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const f32x4 t = *reinterpret_cast<const f32x4 *>(red + (w * 32 + ml) * LDR + nq * 4);
    v += t;
  }
  if (ln_epi) {
    const float mu = rowstat[2 * ml], rstd = rowstat[2 * ml + 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = rstd * (v[e] - mu * cu[e]);
  }
  float xo[4];
  T ob[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = enb + e;
    float x = (v[e] + bi[e]) * sv[e] + rv[e] + av[e];
    x = n < a.N ? apply_act(x, a.act) : 0.f;
    ob[e] = from_f<T>(x);
    xo[e] = a.out_f32 ? x : to_f(ob[e]);
    if (live && n < a.n_store && a.out_f32) static_cast<float *>(a.out)[(size_t)em * a.out_ld + n] = x;
  }
  if (live && !a.out_f32) {
    T *op = static_cast<T *>(a.out) + (size_t)em * a.out_ld + enb;
    if (enb + 3 < a.n_store && (a.out_ld & 3) == 0) __builtin_memcpy(__builtin_assume_aligned(op, 4 * sizeof(T)), ob, 4 * sizeof(T));
    else
      for (int e = 0; e < 4; ++e)
        if (enb + e < a.n_store) op[e] = ob[e];
  }
  if (a.rowpart_out) {   // row-LayerNorm partial of the stored values (ConvGemmArgs, conv_gemm_fast.hip)
    const float mean = sum8_dpp((xo[0] + xo[1]) + (xo[2] + xo[3])) * (1.0f / 32.0f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = xo[e] - mean;
      q = fmaf(d, d, q);
    }
    q = sum8_dpp(q);
    if (nq == 0 && live) *reinterpret_cast<float2 *>(a.rowpart_out + ((size_t)em * a.rowpart_nt + nt) * 2) = make_float2(mean, q);
  }
  if (a.gnpart_out) {   // GroupNorm tile sums of the stored values for the channel-block convolution that follows (kernels.h)
    const float s1 = sum8_dpp((xo[0] + xo[1]) + (xo[2] + xo[3]));
    const float s2 = sum8_dpp(fmaf(xo[0], xo[0], xo[1] * xo[1]) + fmaf(xo[2], xo[2], xo[3] * xo[3]));
    if (nq == 0) {
      gsum[2 * ml] = live ? s1 : 0.f;
      gsum[2 * ml + 1] = live ? s2 : 0.f;
    }
    __syncthreads();
    if (tid < 2) {   // segment 0: rows of the first row's clip; segment 1: rows of the next clip
      const int rbnd = min((m0 / a.Lout + 1) * a.Lout - m0, 32);
      const int lo = tid == 0 ? 0 : rbnd, hi = tid == 0 ? rbnd : 32;
      float t1 = 0.f, t2 = 0.f;
      for (int r = lo; r < hi; ++r) {
        t1 += gsum[2 * r];
        t2 += gsum[2 * r + 1];
      }
      *reinterpret_cast<float2 *>(a.gnpart_out + (((size_t)mt * ntiles + nt) * 2 + tid) * 2) = make_float2(t1, t2);
    }
  }
}

// packed [N][K] (compute type) -> fragment order [N / 32][K / 16][64][8]
template <typename T> __global__ void pack_wfr_kernel(const T *__restrict__ w, int N, int K, T *__restrict__ out) {
  const size_t total = (size_t)N * K;
  const int F = K >> 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const size_t t = e >> 9;
    const int f = (int)(t % F), nt = (int)(t / F);
    out[e] = w[(size_t)(nt * 32 + (lane & 31)) * K + f * 16 + (lane >> 5) * 8 + q];
  }
}

// packed fp32 [N][K] -> split fragment order [N / 32][K / 16][hi | lo'][64][8] (the fp32x engine)
__global__ void pack_wfrx_kernel(const float *__restrict__ w, int N, int K, f16 *__restrict__ out) {
  const size_t total = (size_t)N * K;
  const int F = K >> 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const size_t t = e >> 9;
    const int f = (int)(t % F), nt = (int)(t / F);
    x3_split1<X3_F16>(w[(size_t)(nt * 32 + (lane & 31)) * K + f * 16 + (lane >> 5) * 8 + q], out[t * 1024 + (e & 511)], out[t * 1024 + 512 + (e & 511)]);
  }
}

template <bool CAT> hipError_t launch_rs_x3(const ConvGemmArgs &a, hipStream_t s) {
  const int mtiles = (a.M + 31) / 32, ntiles = (a.n_store + 31) / 32;
  const int swz = (ntiles % 8 == 0) ? 1 : 0;
  const size_t bA = (size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * 4;
  const size_t bA2 = CAT ? (size_t)a.M * a.src2_ld * 4 : 0;
  const size_t bW = (size_t)a.N * a.K * 4;
  const dim3 grid(mtiles * ntiles + (a.pf.ptr && a.pf.bytes >= 16 ? a.pf.wgs : 0));
  const int nf = a.K / 64;
#define SF_RSX(NFD)                                                                                                                                      \
  do {                                                                                                                                                   \
    if (a.taps == 1) hipLaunchKernelGGL((conv_gemm_rs_kernel<float, CAT, NFD, true, true>), grid, dim3(256), 0, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW); \
    else hipLaunchKernelGGL((conv_gemm_rs_kernel<float, CAT, NFD, false, true>), grid, dim3(256), 0, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW); \
  } while (0)
  if (nf <= 8) SF_RSX(8);
  else if (nf <= 12) SF_RSX(12);
  else if (nf <= 16) SF_RSX(16);
  else SF_RSX(20);
#undef SF_RSX
  return hipGetLastError();
}

template <typename T, bool CAT> hipError_t launch_rs(const ConvGemmArgs &a, hipStream_t s) {
  const int mtiles = (a.M + 31) / 32, ntiles = (a.n_store + 31) / 32;
  const int swz = (ntiles % 8 == 0) ? 1 : 0;
  const size_t bA = (size_t)(a.M / a.Lout + (a.M % a.Lout ? 1 : 0)) * a.Lsrc * a.src_ld * 2;
  const size_t bA2 = CAT ? (size_t)a.M * a.src2_ld * 2 : 0;
  const size_t bW = (size_t)a.N * a.K * 2;
  const dim3 grid(mtiles * ntiles + (a.pf.ptr && a.pf.bytes >= 16 ? a.pf.wgs : 0));
  const int nf = a.K / 64;
#define SF_RS(NFD)                                                                                                                                       \
  do {                                                                                                                                                   \
    if (a.taps == 1) hipLaunchKernelGGL((conv_gemm_rs_kernel<T, CAT, NFD, true>), grid, dim3(256), 0, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW); \
    else hipLaunchKernelGGL((conv_gemm_rs_kernel<T, CAT, NFD, false>), grid, dim3(256), 0, s, a, mtiles, ntiles, swz, (unsigned)bA, (unsigned)bA2, (unsigned)bW); \
  } while (0)
  if (nf <= 8) SF_RS(8);
  else if (nf <= 16) SF_RS(16);
  else if (nf <= 24) SF_RS(24);
  else SF_RS(32);
#undef SF_RS
  return hipGetLastError();
}

}  // namespace

// the launch takes the register-staged kernel: fragment-ordered weights at hand, a 16-bit type, 1-D geometry, no prologue, 32x32
// tiles (the caller has decided that), K a multiple of 64 up to 2048, channel counts that are multiples of 16, whole 32-column tiles
bool conv_gemm_rs_ok(int dt, const ConvGemmArgs &a) {
  static const bool off = tune_env("SF_NO_RS") != nullptr;   // A/B aid
  const bool x3 = dt == F32 && a.wfrx != nullptr && a.wx_mode == X3_F16;   // the split-operand form (fp32 activations)
  const size_t es = x3 ? 4 : 2;
  if (off || (dt == F32 && !x3) || (!x3 && !a.wfr) || a.geom != 0 || a.pro != 0 || a.taps < 1) return false;
  if ((a.K % 64) || a.K > (x3 ? 1280 : 2048) || (a.cin % 16) || (a.cin2 % 16) || (a.N % 32) || a.n_store != a.N) return false;
  if (x3 && ((a.src_ld % 4) || (a.cin2 && (a.src2_ld % 4)) || a.out_f32)) return false;   // 16-byte fp32 fragment loads
  if ((a.res && (a.res_ld % 4)) || (a.bscale && (a.bscale_ld % 4)) || (a.badd && (a.badd_ld % 4)) || (a.out_ld % 4)) return false;   // vector epilogue loads
  if (a.ln_ss || a.res_ln) return false;   // the operand-side LayerNorm (Modulation folded into InjectChannels) stays on conv_gemm_fast
  if (a.ln_colsum && (!a.ln_part || a.ln_nt * 32 != a.cin || a.ln_nt > 32 || a.cin2)) return false;
  const size_t lim = 0x7FFFFFF0ull;
  if ((size_t)(a.M / a.Lout + 1) * a.Lsrc * a.src_ld * es >= lim) return false;
  if ((size_t)a.M * (a.src2_ld > 0 ? a.src2_ld : 1) * es >= lim) return false;
  if ((size_t)a.N * a.K * es >= lim) return false;
  return true;
}

hipError_t launch_pack_wfrx(const float *w, int N, int K, void *out, hipStream_t s) {
  if ((N % 32) || (K % 16)) return hipErrorInvalidValue;
  const size_t total = (size_t)N * K;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(pack_wfrx_kernel, dim3(blocks), dim3(256), 0, s, w, N, K, static_cast<f16 *>(out));
  return hipGetLastError();
}

hipError_t launch_conv_gemm_rs(int dt, const ConvGemmArgs &a, hipStream_t s) {
  if (!conv_gemm_rs_ok(dt, a)) return hipErrorInvalidValue;
  if (dt == F32) return a.cin2 ? launch_rs_x3<true>(a, s) : launch_rs_x3<false>(a, s);
  if (dt == BF16) return a.cin2 ? launch_rs<bf16, true>(a, s) : launch_rs<bf16, false>(a, s);
  return a.cin2 ? launch_rs<f16, true>(a, s) : launch_rs<f16, false>(a, s);
}

hipError_t launch_pack_wfr(int dt, const void *w, int N, int K, void *out, hipStream_t s) {
  if (dt == F32 || (N % 32) || (K % 16)) return hipErrorInvalidValue;
  const size_t total = (size_t)N * K;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  if (dt == BF16) hipLaunchKernelGGL((pack_wfr_kernel<bf16>), dim3(blocks), dim3(256), 0, s, static_cast<const bf16 *>(w), N, K, static_cast<bf16 *>(out));
  else hipLaunchKernelGGL((pack_wfr_kernel<f16>), dim3(blocks), dim3(256), 0, s, static_cast<const f16 *>(w), N, K, static_cast<f16 *>(out));
  return hipGetLastError();
}

}  // namespace sf
