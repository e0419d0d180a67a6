// Backward kernels of the ResnetItem building block  y = Conv1d(SiLU(GroupNorm(x)))  and of plain Conv1d (the 1x1 InjectChannels
// conv), fp32, channels-last -- the first slice of the training step (the reference trains in fp32:
// exp/train_diffusion_gh.yaml:87 `precision: 32`, main/module_diffusion.py:73-82).
//
//   forward   u = xhat * gamma + beta,  xhat = (x - mean_g) * rstd_g ;  a = u * sigmoid(u) ;  y[b,l,n] = sum_{t,c} W[n][c][t] a[b, l+t-pad, c] + bias[n]
//   dgrad     da[b,l,c] = sum_{t,n} dy[b, l-t+pad, n] W[n][c][t]      -> the FORWARD implicit-GEMM kernels on flipped / transposed weights
//   wgrad     dW[n][c][t] = sum_{b,l} dy[b,l,n] a[b, l+t-pad, c]       -> conv_wgrad_kernel: a "TN" GEMM whose reduction runs over the rows;
//             fp32 MFMA 32x32x2 takes both operands with the reduction index across the two half-waves, so dy and a are read
//             straight from their row-major layout (32 consecutive floats per half-wave); split over row ranges, summed in a fixed order
//   db[n]     = sum_{b,l} dy[b,l,n]                                      -> col_sums_kernel (two deterministic stages)
//   GroupNorm+SiLU backward (gn_silu_bwd_kernel, one workgroup per (clip, group)):
//             du = da * sigmoid(u) (1 + u (1 - sigmoid(u)));  g = du * gamma;  dx = rstd (g - mean(g) - xhat mean(g xhat));
//             dgamma[c] = sum du xhat,  dbeta[c] = sum du   (per-clip partials, reduced over clips afterwards)
// No atomics anywhere: gradients are bit-reproducible.
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

// Conv1d weight (N, C, taps) -> dgrad weight matrix [c][t' * ldn + n] = W[n][c][taps-1-t']  (rows padded to ldn columns per tap)
__global__ void pack_dgrad_kernel(const float *__restrict__ w, int N, int C, int taps, int ldn, float *__restrict__ out, bf16 *__restrict__ outx) {
  const int64_t total = (int64_t)C * taps * ldn;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % ldn);
    const int64_t r = i / ldn;
    const int t = (int)(r % taps), c = (int)(r / taps);
    const float v = n < N ? w[((int64_t)n * C + c) * taps + (taps - 1 - t)] : 0.f;
    out[i] = v;
    if (outx) x3_split1<X3_BF16>(v, outx[(i >> 5) * 64 + (i & 31)], outx[(i >> 5) * 64 + 32 + (i & 31)]);   // split bf16 image, same pass
  }
}

// Every weight image one training convolution reads, in ONE launch per weight and step (the forward and the data-gradient GEMM used to pack
// their own: 2 x 200 launches per step, the data-gradient one with reads strided by C * taps floats).  A workgroup takes a 32 (n) x 32 (c)
// tile with all its taps through LDS: the (c, tap) run of a row is contiguous in the PyTorch layout, the forward images are written with c
// fastest, the data-gradient images with n fastest.  Any of the four outputs may be null:
//   fw  [n][tap][c] fp32                      fwx  its split fp16 image (hi | lo' per 32 columns; (taps * C) % 32 == 0)
//   dg  [c][taps - 1 - tap][n] fp32            dgx  its split bf16 image ((taps * N) % 32 == 0)
constexpr int PT_MAX_TAPS = 9;
template <int TP /* compile-time tap count (1, 3), 0 = any up to PT_MAX_TAPS */>
__device__ __forceinline__ void pack_train_tile(const float *__restrict__ w, int N, int C, int taps_rt, float *__restrict__ fw, f16 *__restrict__ fwx,
                                                float *__restrict__ dg, bf16 *__restrict__ dgx, const int c0, const int n0, float *tile) {
  const int taps = TP ? TP : taps_rt;
  const int cw = min(32, C - c0), nh = min(32, N - n0);
  const int run = cw * taps;   // floats of one row's (c, tap) run inside the tile
  for (int e = threadIdx.x; e < nh * run; e += 256) {
    const int nl = e / run, rem = e - nl * run;
    const int cl = rem / taps, tap = rem - cl * taps;
    tile[(tap * 32 + nl) * 33 + cl] = w[((int64_t)(n0 + nl) * C + c0) * taps + rem];
  }
  __syncthreads();
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;   // 8 groups of 32 lanes
  if (fw || fwx) {
    for (int r = grp; r < taps * nh; r += 8) {   // r -> (tap, nl); lanes along c
      const int tap = r / nh, nl = r - tap * nh;
      if (lane < cw) {
        const float v = tile[(tap * 32 + nl) * 33 + lane];
        const int64_t i = ((int64_t)(n0 + nl) * taps + tap) * C + c0 + lane;
        if (fw) fw[i] = v;
        if (fwx) x3_split1<X3_F16>(v, fwx[(i >> 5) * 64 + (i & 31)], fwx[(i >> 5) * 64 + 32 + (i & 31)]);
      }
    }
  }
  if (dg || dgx) {
    for (int r = grp; r < taps * cw; r += 8) {   // r -> (tap, cl); lanes along n
      const int tap = r / cw, cl = r - tap * cw;
      if (lane < nh) {
        const float v = tile[(tap * 32 + lane) * 33 + cl];
        const int64_t j = ((int64_t)(c0 + cl) * taps + (taps - 1 - tap)) * N + n0 + lane;
        if (dg) dg[j] = v;
        if (dgx) x3_split1<X3_BF16>(v, dgx[(j >> 5) * 64 + (j & 31)], dgx[(j >> 5) * 64 + 32 + (j & 31)]);
      }
    }
  }
}
template <int TP>
__global__ __launch_bounds__(256) void pack_train_kernel(const float *__restrict__ w, int N, int C, int taps_rt, float *__restrict__ fw, f16 *__restrict__ fwx,
                                                         float *__restrict__ dg, bf16 *__restrict__ dgx) {
  __shared__ float tile[(TP ? TP : PT_MAX_TAPS) * 32 * 33];
  pack_train_tile<TP>(w, N, C, taps_rt, fw, fwx, dg, dgx, blockIdx.x * 32, blockIdx.y * 32, tile);
}

// The same for MANY weights in one launch (the training step's whole weight set: 240 launches of ~8 us were mostly launch latency).  The
// descriptor table lives in device memory, 7 64-bit words per weight: w, fw, fwx, dg, dgx (addresses; 0 = skipped), N | C << 32,
// taps | first_tile << 32; a workgroup finds its weight by bisection over first_tile (items are sorted by it).
constexpr int PACK_DESC_WORDS = 7;
__global__ __launch_bounds__(256) void pack_train_many_kernel(const unsigned long long *__restrict__ desc, const int n_items) {
  __shared__ float tile[PT_MAX_TAPS * 32 * 33];
  const unsigned blk = blockIdx.x;
  int lo = 0, hi = n_items - 1;   // the last item whose first_tile <= blk (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((unsigned)(desc[mid * PACK_DESC_WORDS + 6] >> 32) <= blk) lo = mid;
    else hi = mid - 1;
  }
  const unsigned long long *d = desc + lo * PACK_DESC_WORDS;
  const float *w = reinterpret_cast<const float *>(d[0]);
  float *fw = reinterpret_cast<float *>(d[1]);
  f16 *fwx = reinterpret_cast<f16 *>(d[2]);
  float *dg = reinterpret_cast<float *>(d[3]);
  bf16 *dgx = reinterpret_cast<bf16 *>(d[4]);
  const int N = (int)(d[5] & 0xffffffffu), C = (int)(d[5] >> 32), taps = (int)(d[6] & 0xffffffffu);
  const int t = (int)(blk - (unsigned)(d[6] >> 32)), ntx = (C + 31) / 32;
  const int c0 = (t % ntx) * 32, n0 = (t / ntx) * 32;
  if (n0 >= N) return;
  if (taps == 1) pack_train_tile<1>(w, N, C, taps, fw, fwx, dg, dgx, c0, n0, tile);
  else if (taps == 3) pack_train_tile<3>(w, N, C, taps, fw, fwx, dg, dgx, c0, n0, tile);
  else pack_train_tile<0>(w, N, C, taps, fw, fwx, dg, dgx, c0, n0, tile);
}

// partial[s][n][q],  q = t * C + c:  sum over the rows of split s of dy[row][n] * a[row + t - pad][c]
// A wave holds TN x TQ accumulator tiles of 32 x 32 (2 x 2 for the wide layers: one A and one B value per tile row / column feed
// TN * TQ MFMAs, which halves the L2 traffic per flop); the 4 waves take interleaved row pairs and are summed through LDS.
template <int TN, int TQ>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ act, int rows, int L, int C, int N,
                                                         int taps, int pad, int rows_per_split, float *__restrict__ partial,
                                                         float *__restrict__ dw_direct /* with ONE split: PyTorch layout, no reduce pass */) {
  __shared__ float red[4][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int Q = taps * C;
  int n[TN], c[TQ], shift[TQ];
  bool qok[TQ];
#pragma unroll
  for (int i = 0; i < TN; ++i) n[i] = (blockIdx.x * TN + i) * 32 + fr;   // A operand: lane -> output channel
#pragma unroll
  for (int j = 0; j < TQ; ++j) {                                         // B operand: lane -> (tap, input channel)
    const int q = (blockIdx.y * TQ + j) * 32 + fr;
    qok[j] = q < Q;
    const int t = qok[j] ? q / C : 0;
    c[j] = qok[j] ? q - t * C : 0;
    shift[j] = t - pad;
  }
  const int r_begin = blockIdx.z * rows_per_split, r_end = min(rows, r_begin + rows_per_split);
  f32x16 acc[TN][TQ];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // wave w takes the row pairs w, w + 4, ...; inside a pair the half-wave picks the row (k index of the 32x32x2 MFMA).  The
  // operands of UNR row pairs are fetched before the first MFMA of the group, so UNR * (TN + TQ) loads are in flight per wave.
  // (16 pairs for the single-tile form: a wave of the 8-channel level walks 256 row pairs, and with 4 pairs in flight it spent 64
  //  memory round trips doing so)
  constexpr int UNR = TN * TQ == 1 ? 16 : (TN * TQ <= 3 ? 8 : 4);
  // this lane's row walks r, r + 8, ...: its position inside the clip and its operand addresses advance incrementally (the first
  // version recomputed r % L and two 64-bit products per fetch: ~80 vector instructions per row pair, the thin levels' whole cost)
  int r = r_begin + 2 * wave + fh;
  int l = r % L;
  const float *pa[TN];
  const float *pb[TQ];
  bool nok[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    nok[i] = n[i] < N;
    pa[i] = dy + (size_t)r * N + (nok[i] ? n[i] : 0);
  }
#pragma unroll
  for (int j = 0; j < TQ; ++j) pb[j] = act + ((ptrdiff_t)r + shift[j]) * C + c[j];
  const size_t stepA = (size_t)8 * N, stepB = (size_t)8 * C;
  for (int r0 = r_begin + 2 * wave; r0 < r_end; r0 += 8 * UNR) {
    float av[UNR][TN], bv[UNR][TQ];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {   // rows past r_end load zeros
      const bool rv = r < r_end;
#pragma unroll
      for (int i = 0; i < TN; ++i) av[u][i] = (rv && nok[i]) ? *pa[i] : 0.f;
#pragma unroll
      for (int j = 0; j < TQ; ++j) bv[u][j] = (rv && qok[j] && (unsigned)(l + shift[j]) < (unsigned)L) ? *pb[j] : 0.f;
      r += 8;
      l += 8;
      if (l >= L) l %= L;
#pragma unroll
      for (int i = 0; i < TN; ++i) pa[i] += stepA;
#pragma unroll
      for (int j = 0; j < TQ; ++j) pb[j] += stepB;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      if (i + j) __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) red[wave][(e & 3) + 8 * (e >> 2) + 4 * fh][fr] = acc[i][j][e];
      __syncthreads();
      for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
        const int a = idx >> 5, b = idx & 31;
        const float v = (red[0][a][b] + red[1][a][b]) + (red[2][a][b] + red[3][a][b]);
        const int nn = (blockIdx.x * TN + i) * 32 + a, qq = (blockIdx.y * TQ + j) * 32 + b;
        if (nn < N && qq < Q) {
          if (dw_direct) {
            const int t = qq / C, cc = qq - t * C;
            dw_direct[((size_t)nn * C + cc) * taps + t] = v;
          } else {
            partial[((size_t)blockIdx.z * N + nn) * Q + qq] = v;
          }
        }
      }
    }
}

// The same product for the wide layers (N >= 64, Q >= 64, C % 32 == 0) with the operands staged in LDS: a workgroup owns a
// (64 TW) x (64 TW) tile of dw (4 waves as 2 x 2, TW x TW accumulator tiles each) and walks its row range in chunks of 32 rows;
// dy rows and the activation rows the chunk's taps touch are fetched once per workgroup with 16-byte loads (the next chunk's
// loads are in flight while this one is multiplied) -- 32 (TW = 2) flops per byte of L2 traffic instead of 16, and no LDS
// reduction at the end (every wave owns its outputs).  Two staging modes:
//   single tap (taps == 1, or C a multiple of the tile width): the tile's columns are one tap's channels [cw0, cw0 + 64 TW);
//   whole rows (C <= 64 TW): all C channels of the chunk's rows plus the halo rows of the other taps.
// Rows whose shifted position leaves the clip are masked when the B operand is read (lrow[] = position inside the clip).
template <int TW>
__global__ __launch_bounds__(256) void conv_wgrad_lds_kernel(const float *__restrict__ dy, const float *__restrict__ act, int rows, int L, int C, int N,
                                                             int taps, int pad, int rows_per_split, float *__restrict__ partial,
                                                             float *__restrict__ dw_direct) {
  constexpr int KR = 32, TILE = 64 * TW, PITCH = TILE + 32, NRMAX = KR + 8;
  __shared__ __attribute__((aligned(16))) float dyS[KR * PITCH];
  __shared__ __attribute__((aligned(16))) float actS[NRMAX * PITCH];
  __shared__ int lrow[KR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
  const int wn = wave >> 1, wq = wave & 1;
  const int Q = taps * C;
  const int n0 = blockIdx.x * TILE, q0 = blockIdx.y * TILE;
  const bool single = taps == 1 || (C % TILE) == 0;
  const int t0 = single ? q0 / C : 0;
  const int cw0 = single ? q0 - t0 * C : 0;                 // first staged channel
  const int cwid = single ? min(TILE, C - cw0) : C;          // staged channels per row
  const int nr = single ? KR : KR + taps - 1;                // staged rows per chunk
  const int rshift = single ? t0 - pad : -pad;               // staged row k' is global row r0 + k' + rshift
  // this wave's column sub-tiles: tap (for the clip mask), row offset and channel offset inside the staged window
  int tj[TW], koff[TW], coff[TW];
  bool qok[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    const int qs = q0 + (wq * TW + j) * 32;
    qok[j] = qs < Q;
    const int t = qok[j] ? qs / C : 0;
    tj[j] = t;
    koff[j] = single ? 0 : t;
    coff[j] = single ? (wq * TW + j) * 32 : (qok[j] ? qs - t * C : 0);
  }
  const int r_begin = blockIdx.z * rows_per_split, r_end = min(rows, r_begin + rows_per_split);
  f32x16 acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int DYV = KR * TILE / 4 / 256;                   // 16-byte loads of dy per thread and chunk
  constexpr int ACV = (NRMAX * TILE / 4 + 255) / 256;        // upper bound for the activation window
  f32x4 pdy[DYV], pac[ACV];
  const int vpr = cwid / 4;                                  // 16-byte vectors per staged activation row (C % 32 == 0)
  auto fetch = [&](int r0) {
#pragma unroll
    for (int u = 0; u < DYV; ++u) {
      const int idx = tid + 256 * u, k = idx / (TILE / 4), c4 = idx - k * (TILE / 4);
      const int r = r0 + k, n = n0 + 4 * c4;
      pdy[u] = (r < r_end && n < N) ? *reinterpret_cast<const f32x4 *>(dy + (size_t)r * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < ACV; ++u) {
      const int idx = tid + 256 * u, k = idx / vpr, c4 = idx - k * vpr;
      const int g = r0 + k + rshift;
      pac[u] = (k < nr && g >= 0 && g < rows) ? *reinterpret_cast<const f32x4 *>(act + (size_t)g * C + cw0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&](int r0) {
#pragma unroll
    for (int u = 0; u < DYV; ++u) {
      const int idx = tid + 256 * u, k = idx / (TILE / 4), c4 = idx - k * (TILE / 4);
      *reinterpret_cast<f32x4 *>(dyS + k * PITCH + 4 * c4) = pdy[u];
    }
#pragma unroll
    for (int u = 0; u < ACV; ++u) {
      const int idx = tid + 256 * u, k = idx / vpr, c4 = idx - k * vpr;
      if (k < nr) *reinterpret_cast<f32x4 *>(actS + k * PITCH + 4 * c4) = pac[u];
    }
    if (tid < KR) lrow[tid] = (r0 + tid) % L;
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += KR) {
    __syncthreads();
    stash(r0);
    __syncthreads();
    if (r0 + KR < r_end) fetch(r0 + KR);
#pragma unroll 4
    for (int st = 0; st < KR / 2; ++st) {
      const int k = 2 * st + fh;
      const int l = lrow[k];
      float av[TW], bv[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) av[i] = dyS[k * PITCH + (wn * TW + i) * 32 + fr];
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const float v = actS[(k + koff[j]) * PITCH + coff[j] + fr];
        bv[j] = (qok[j] && (unsigned)(l + tj[j] - pad) < (unsigned)L) ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      const int qq = q0 + (wq * TW + j) * 32 + fr;
      if (qq >= Q) continue;
      const int t = qq / C, cc = qq - t * C;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nn = n0 + (wn * TW + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (nn < N) {
          if (dw_direct) dw_direct[((size_t)nn * C + cc) * taps + t] = acc[i][j][e];
          else partial[((size_t)blockIdx.z * N + nn) * Q + qq] = acc[i][j][e];
        }
      }
    }
}

// The LDS-staged weight gradient on split fp16 operands (SF_F32X; common.h, x3_split): both operands are activations, so they are split
// ONCE per element while the chunk is staged -- dy and the activation window land in LDS as (hi, lo') fp16 images, row-major
// [row][column] -- and the k-strided fragments (8 consecutive ROWS of one column per lane) are gathered by the hardware transpose
// read `ds_read_b64_tr_b16` (cdna_hip_programming.md T10): per 16-lane group a 4-row x 16-column block, column i to lane i.
// A chunk of 32 rows = two 16-deep products per accumulator tile, three MFMAs each.  Rows whose shifted position leaves the clip are
// cleared in the fragment through per-tap validity masks of the chunk (all ones except at clip boundaries: wave-uniform fast path).
typedef short v4i16_w __attribute__((ext_vector_type(4)));
typedef short v8i16_w __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) v4i16_w lds_v4i16_w;

template <int TW, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_x3_kernel(const float *__restrict__ dy, const float *__restrict__ act, int rows, int L, int C, int N,
                                                            int taps, int pad, int rows_per_split, float *__restrict__ partial,
                                                            float *__restrict__ dw_direct) {
  // pitch in fp16 elements: consecutive rows 64 B apart modulo 256 B, so the 4 rows x 2 column groups a 32-lane half gathers with one
  // transposed read fall into eight disjoint 8-bank ranges
  constexpr int KR = 32, TILE = 64 * TW, PITCH = TILE + 32, NRMAX = KR + 8;
  using XE = typename X3P<MODE>::elem;
  using XV = typename X3P<MODE>::v8;
  __shared__ __attribute__((aligned(16))) XE dyH[KR * PITCH], dyL[KR * PITCH];
  __shared__ __attribute__((aligned(16))) XE acH[NRMAX * PITCH], acL[NRMAX * PITCH];
  __shared__ unsigned vmask[9];   // bit k of vmask[t]: row k of the chunk may read tap t (its shifted position stays inside the clip)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
  const int wn = wave >> 1, wq = wave & 1;
  const int Q = taps * C;
  const int n0 = blockIdx.x * TILE, q0 = blockIdx.y * TILE;
  const bool single = taps == 1 || (C % TILE) == 0;
  const int t0 = single ? q0 / C : 0;
  const int cw0 = single ? q0 - t0 * C : 0;
  const int cwid = single ? min(TILE, C - cw0) : C;
  const int nr = single ? KR : KR + taps - 1;
  const int rshift = single ? t0 - pad : -pad;
  int tj[TW], koff[TW], coff[TW];
  bool qok[TW];
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    const int qs = q0 + (wq * TW + j) * 32;
    qok[j] = qs < Q;
    const int t = qok[j] ? qs / C : 0;
    tj[j] = t;
    koff[j] = single ? 0 : t;
    coff[j] = single ? (wq * TW + j) * 32 : (qok[j] ? qs - t * C : 0);
  }
  const int r_begin = blockIdx.z * rows_per_split, r_end = min(rows, r_begin + rows_per_split);
  constexpr bool ONE = MODE == X3_BF16;   // no scale between the parts: one accumulator per tile
  f32x16 acc[TW][TW], accL[ONE ? 1 : TW][ONE ? 1 : TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if constexpr (!ONE) accL[i][j][e] = 0.f;
      }
  constexpr int DYV = KR * TILE / 4 / 256;
  constexpr int ACV = (NRMAX * TILE / 4 + 255) / 256;
  f32x4 pdy[DYV], pac[ACV];
  const int vpr = cwid / 4;
  auto fetch = [&](int r0) {
#pragma unroll
    for (int u = 0; u < DYV; ++u) {
      const int idx = tid + 256 * u, k = idx / (TILE / 4), c4 = idx - k * (TILE / 4);
      const int r = r0 + k, n = n0 + 4 * c4;
      pdy[u] = (r < r_end && n < N) ? *reinterpret_cast<const f32x4 *>(dy + (size_t)r * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < ACV; ++u) {
      const int idx = tid + 256 * u, k = idx / vpr, c4 = idx - k * vpr;
      const int g = r0 + k + rshift;
      pac[u] = (k < nr && g >= 0 && g < rows) ? *reinterpret_cast<const f32x4 *>(act + (size_t)g * C + cw0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  typedef XE f16x4 __attribute__((ext_vector_type(4)));
  auto split4 = [](const f32x4 v, f16x4 &hi, f16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      XE h, l;
      x3_split1<MODE>(v[e], h, l);
      hi[e] = h;
      lo[e] = l;
    }
  };
  auto stash = [&](int r0) {
#pragma unroll
    for (int u = 0; u < DYV; ++u) {
      const int idx = tid + 256 * u, k = idx / (TILE / 4), c4 = idx - k * (TILE / 4);
      f16x4 hi, lo;
      split4(pdy[u], hi, lo);
      *reinterpret_cast<f16x4 *>(dyH + k * PITCH + 4 * c4) = hi;
      *reinterpret_cast<f16x4 *>(dyL + k * PITCH + 4 * c4) = lo;
    }
#pragma unroll
    for (int u = 0; u < ACV; ++u) {
      const int idx = tid + 256 * u, k = idx / vpr, c4 = idx - k * vpr;
      if (k < nr) {
        f16x4 hi, lo;
        split4(pac[u], hi, lo);
        *reinterpret_cast<f16x4 *>(acH + k * PITCH + 4 * c4) = hi;
        *reinterpret_cast<f16x4 *>(acL + k * PITCH + 4 * c4) = lo;
      }
    }
    if (wave == 0) {   // validity masks of the chunk: one ballot per tap over the 32 rows (lanes 32-63 repeat them)
      const int l = (r0 + (lane & 31)) % L;
      for (int t = 0; t < taps; ++t) {
        const unsigned long long b = __ballot((unsigned)(l + t - pad) < (unsigned)L);
        if (lane == 0) vmask[t] = (unsigned)b;
      }
    }
  };
  // transposed-read lane geometry: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of the group's 16 columns
  const int tr_off = (8 * fh + ((lane & 15) >> 2)) * PITCH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  auto trfrag = [&](const XE *base) {
    const v4i16_w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_w *)(base));
    const v4i16_w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_w *)(base + 4 * PITCH));
    const v8i16_w both = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return both;
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += KR) {
    __syncthreads();
    stash(r0);
    __syncthreads();
    if (r0 + KR < r_end) fetch(r0 + KR);
    unsigned vm[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j) vm[j] = vmask[tj[j]];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      XV ah[TW], al[TW], bh[TW], bl[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        const int o = (16 * st) * PITCH + (wn * TW + i) * 32 + tr_off;
        ah[i] = __builtin_bit_cast(XV, trfrag(dyH + o));
        al[i] = __builtin_bit_cast(XV, trfrag(dyL + o));
      }
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const int o = (16 * st + koff[j]) * PITCH + coff[j] + tr_off;
        v8i16_w h = trfrag(acH + o), l = trfrag(acL + o);
        if (vm[j] != 0xffffffffu) {   // wave-uniform: a clip boundary inside the chunk -- clear the rows that may not read this tap
          const unsigned bits = vm[j] >> (16 * st + 8 * fh);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const short keep = (bits >> e) & 1u ? (short)-1 : (short)0;
            h[e] &= keep;
            l[e] &= keep;
          }
        }
        bh[j] = __builtin_bit_cast(XV, h);
        bl[j] = __builtin_bit_cast(XV, l);
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
          if constexpr (ONE) x3_mfma1_bf16(ah[i], al[i], bh[j], bl[j], acc[i][j]);
          else x3_mfma<MODE>(ah[i], al[i], bh[j], bl[j], acc[i][j], accL[i][j]);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      const int qq = q0 + (wq * TW + j) * 32 + fr;
      if (qq >= Q) continue;
      const int t = qq / C, cc = qq - t * C;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nn = n0 + (wn * TW + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (nn < N) {
          float v = acc[i][j][e];
          if constexpr (!ONE) v = fmaf(accL[i][j][e], X3P<MODE>::INV, v);
          if (dw_direct) dw_direct[((size_t)nn * C + cc) * taps + t] = v;
          else partial[((size_t)blockIdx.z * N + nn) * Q + qq] = v;
        }
      }
    }
}

// Sum over the leading (slice) dimension: 32 outputs per workgroup, 8 threads per output take the slices k = kq, kq + 8, ... and
// are combined through LDS in a fixed order (deterministic; the loads of a 32-lane group are 128 contiguous bytes).
__device__ __forceinline__ float slice_sum_8(const float *__restrict__ part, int S, size_t stride, size_t col, bool valid, float *sh /* [256] */) {
  const int kq = threadIdx.x >> 5;
  float s = 0.f;
  if (valid) {
#pragma unroll 8
    for (int k = kq; k < S; k += 8) s += part[(size_t)k * stride + col];   // (eight loads in flight per thread; the sum keeps its order)
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  float t = 0.f;
  if (kq == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) t += sh[j * 32 + threadIdx.x];
  }
  return t;
}

// A slice reduction riding on another launch: out[j] = sum_k part[k][j], 32 outputs per workgroup, done by the workgroups the host kernel
// appends to its grid (the small reductions of the backward pass were 2 launches each: 250 launches of ~6 us per training step).
struct TailReduce {
  const float *part = nullptr;
  float *out = nullptr;
  int S = 0, cols = 0;
  __host__ __device__ int blocks() const { return part ? (cols + 31) / 32 : 0; }
};
__device__ __forceinline__ void tail_reduce_block(const TailReduce &t, int block, float *sh /* [256] */) {
  const int col = block * 32 + (threadIdx.x & 31);
  const bool valid = col < t.cols;
  const float v = slice_sum_8(t.part, t.S, (size_t)t.cols, (size_t)(valid ? col : 0), valid, sh);
  if (threadIdx.x < 32 && valid) t.out[col] = v;
}

// dw[n][c][t] (PyTorch layout) = sum_s partial[s][n][t * C + c]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int S, int N, int C, int taps, float *__restrict__ dw,
                                                           const int nb_main, const TailReduce tail /* the bias gradient's slices */) {
  __shared__ float sh[256];
  if ((int)blockIdx.x >= nb_main) {
    tail_reduce_block(tail, (int)blockIdx.x - nb_main, sh);
    return;
  }
  const int64_t total = (int64_t)N * C * taps;
  const int Q = taps * C;
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
  const bool valid = i < total;
  const int64_t ii = valid ? i : 0;
  const int t = (int)(ii % taps);
  const int64_t r = ii / taps;
  const int c = (int)(r % C), n = (int)(r / C);
  const float v = slice_sum_8(partial, S, (size_t)N * Q, (size_t)n * Q + (size_t)t * C + c, valid, sh);
  if (threadIdx.x < 32 && valid) dw[i] = v;
}

// part[blockIdx.y][col] = sum of x[row][col] over the rows of the slice (one thread per column, rows strided by gridDim.y)
__global__ void col_sums_kernel(const float *__restrict__ x, int64_t rows, int cols, int64_t rows_per_slice, float *__restrict__ part) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += x[r * cols + col];
  part[(size_t)blockIdx.y * cols + col] = s;
}
// the same with V (4 or 1) columns per access, (cols / V) dividing 256: a thread always meets the same V columns; the
// 256 / (cols / V) threads that share a column set are summed in thread order through LDS
template <int V>
__global__ __launch_bounds__(256) void col_sums_vec_kernel(const float *__restrict__ x, int64_t rows, int cols, int64_t rows_per_slice,
                                                           float *__restrict__ part) {
  __shared__ float red[V * 256];
  const int tid = threadIdx.x, vpr = cols / V, cv = tid % vpr, rstep = 256 / vpr;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int64_t r = r0 + tid / vpr; r < r1; r += rstep) {
    if constexpr (V == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(x + r * cols + 4 * cv);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += v[j];
    } else {
      acc[0] += x[r * cols + cv];
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[j * 256 + tid] = acc[j];
  __syncthreads();
  for (int c = tid; c < cols; c += 256) {
    const float *rp = red + (c % V) * 256 + (c / V);
    float t = 0.f;
    for (int sl = 0; sl < rstep; ++sl) t += rp[sl * vpr];
    part[(size_t)blockIdx.x * cols + c] = t;
  }
}
// the same for C % 4 == 0: a thread owns four consecutive channels of one (n, tap) and walks the slices with 16-byte loads along the
// slabs' contiguous dimension (the kernel above gathers 4-byte elements C apart: 3.5 ms of a 76 ms training step); slices are added in
// index order, four partial sums deep (fixed order: bit-reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_vec_kernel(const float *__restrict__ partial, int S, int N, int C, int taps, float *__restrict__ dw,
                                                               const int nb_main, const TailReduce tail /* the bias gradient's slices */) {
  __shared__ float sh[256];
  if ((int)blockIdx.x >= nb_main) {
    tail_reduce_block(tail, (int)blockIdx.x - nb_main, sh);
    return;
  }
  const int Q = taps * C;
  const int64_t NQ = (int64_t)N * Q;
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= NQ) return;
  const int n = (int)(e / Q), q = (int)(e - (int64_t)n * Q);
  const int t = q / C, c = q - t * C;
  const float *p = partial + e;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int k = 0;
  for (; k + 8 <= S; k += 8) {   // eight slices per memory round trip (the partial sums keep their four interleaved chains: same bits)
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (size_t)(k + u) * NQ);
    a0 += v[0];
    a1 += v[1];
    a2 += v[2];
    a3 += v[3];
    a0 += v[4];
    a1 += v[5];
    a2 += v[6];
    a3 += v[7];
  }
  for (; k + 4 <= S; k += 4) {
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(p + (size_t)k * NQ);
    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(p + (size_t)(k + 1) * NQ);
    const f32x4 v2 = *reinterpret_cast<const f32x4 *>(p + (size_t)(k + 2) * NQ);
    const f32x4 v3 = *reinterpret_cast<const f32x4 *>(p + (size_t)(k + 3) * NQ);
    a0 += v0;
    a1 += v1;
    a2 += v2;
    a3 += v3;
  }
  for (; k < S; ++k) a0 += *reinterpret_cast<const f32x4 *>(p + (size_t)k * NQ);
  const f32x4 r = (a0 + a1) + (a2 + a3);
  float *o = dw + ((size_t)n * C + c) * taps + t;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[(size_t)j * taps] = r[j];
}
// out[j] = sum_k part[k][j]
__global__ __launch_bounds__(256) void slices_reduce_kernel(const float *__restrict__ part, int S, int cols, float *__restrict__ out) {
  __shared__ float sh[256];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31);
  const bool valid = col < cols;
  const float v = slice_sum_8(part, S, (size_t)cols, (size_t)(valid ? col : 0), valid, sh);
  if (threadIdx.x < 32 && valid) out[col] = v;
}

__device__ __forceinline__ float block_sum_256(float v, float *sh) {   // every thread gets the total; fixed order
  v = wave_sum_dpp(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One workgroup per (clip, group).  cpg must divide 256 (a thread then always meets the same channel).
__global__ __launch_bounds__(256) void gn_silu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ da, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, int L, int C, int G, float eps, float *__restrict__ dx,
                                                          float *__restrict__ dgb_part /* [B][2][C] */, const float *__restrict__ dx_add /* or null */) {
  __shared__ float sh[4];
  __shared__ float chs[2][256];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int cpg = C / G;
  const int n_el = L * cpg;
  const float inv_n = 1.0f / (float)n_el;
  const float *xb = x + (size_t)b * L * C + g * cpg;
  const float *db = da + (size_t)b * L * C + g * cpg;
  float *ob = dx + (size_t)b * L * C + g * cpg;
  const float *ab = dx_add ? dx_add + (size_t)b * L * C + g * cpg : nullptr;
  const int tc = threadIdx.x % cpg;   // this thread's channel inside the group (256 % cpg == 0)
  float s = 0.f;
  for (int i = threadIdx.x; i < n_el; i += 256) s += xb[(size_t)(i / cpg) * C + (i % cpg)];
  const float mean = block_sum_256(s, sh) * inv_n;
  float sq = 0.f;
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const float d = xb[(size_t)(i / cpg) * C + (i % cpg)] - mean;
    sq = fmaf(d, d, sq);
  }
  const float rstd = rsqrtf(block_sum_256(sq, sh) * inv_n + eps);
  const float gam = gamma[g * cpg + tc], bet = beta[g * cpg + tc];
  float s1 = 0.f, s2 = 0.f, dgam = 0.f, dbet = 0.f;
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const size_t off = (size_t)(i / cpg) * C + (i % cpg);
    const float xh = (xb[off] - mean) * rstd;
    const float u = fmaf(xh, gam, bet);
    const float sg = 1.0f / (1.0f + expf(-u));
    const float du = db[off] * sg * (1.0f + u * (1.0f - sg));
    const float g1 = du * gam;
    s1 += g1;
    s2 = fmaf(g1, xh, s2);
    dgam = fmaf(du, xh, dgam);
    dbet += du;
  }
  const float m1 = block_sum_256(s1, sh) * inv_n, m2 = block_sum_256(s2, sh) * inv_n;
  chs[0][threadIdx.x] = dgam;
  chs[1][threadIdx.x] = dbet;
  __syncthreads();
  if ((int)threadIdx.x < cpg) {   // fixed-order sum over the threads that share this channel
    float a0 = 0.f, a1 = 0.f;
    for (int k = threadIdx.x; k < 256; k += cpg) {
      a0 += chs[0][k];
      a1 += chs[1][k];
    }
    dgb_part[((size_t)b * 2 + 0) * C + g * cpg + threadIdx.x] = a0;
    dgb_part[((size_t)b * 2 + 1) * C + g * cpg + threadIdx.x] = a1;
  }
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const size_t off = (size_t)(i / cpg) * C + (i % cpg);
    const float xh = (xb[off] - mean) * rstd;
    const float u = fmaf(xh, gam, bet);
    const float sg = 1.0f / (1.0f + expf(-u));
    const float g1 = db[off] * sg * (1.0f + u * (1.0f - sg)) * gam;
    ob[off] = rstd * (g1 - m1 - xh * m2) + (ab ? ab[off] : 0.f);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// LayerNorm-modulate backward:  y = xhat (1 + s_b) + t_b,  xhat = LayerNorm_C(x; eps, no affine)
//   dx = rstd (g - mean_c(g) - xhat mean_c(g xhat)),  g = dy (1 + s_b);   ds_b[c] = sum_l dy xhat,  dt_b[c] = sum_l dy
// One workgroup per (clip, chunk of rows).  The per-clip sums are accumulated per lane over its rows, reduced over the row groups
// through LDS and written per chunk; chunks_reduce_kernel adds the chunks (fixed order).
// ---------------------------------------------------------------------------------------------------------------------------
// C = 4 * TPR * PER (the powers of two from 4 to 1024 -- what the forward kernel accepts): TPR lanes share a row (16-byte accesses,
// PER of them per lane), so a wave covers 64 / TPR rows at once and narrow levels keep every lane busy; the row sums are shuffles
// inside the TPR-lane group.
template <int TPR, int PER>
__global__ __launch_bounds__(256) void ln_mod_bwd_vec_kernel(const float *__restrict__ x, const float *__restrict__ ss, int ss_ld,
                                                             const float *__restrict__ dy, int L, float eps, int rows_per_chunk, float *__restrict__ dx,
                                                             float *__restrict__ dss_part /* [B][nchunk][2C] */, const float *__restrict__ dx_add /* or null */) {
  constexpr int C = 4 * TPR * PER, NG = 256 / TPR;
  __shared__ float red[2 * NG * C];
  const int tid = threadIdx.x, sub = tid % TPR, grp = tid / TPR;
  const int b = blockIdx.x, chunk = blockIdx.y, nchunk = gridDim.y;
  float sc[PER][4], acs[PER][4], act[PER][4];
#pragma unroll
  for (int k = 0; k < PER; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[k][j] = 1.0f + (ss ? ss[(size_t)b * ss_ld + 4 * (sub + TPR * k) + j] : 0.f);
      acs[k][j] = act[k][j] = 0.f;
    }
  const int l0 = chunk * rows_per_chunk, l1 = min(L, l0 + rows_per_chunk);
  constexpr float inv_c = 1.0f / (float)C;
  auto group_sum = [](float v) {
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, TPR);
    return v;
  };
  for (int l = l0 + grp; l < l1; l += NG) {
    const size_t row = ((size_t)b * L + l) * C;
    f32x4 xv[PER], dv[PER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      xv[k] = *reinterpret_cast<const f32x4 *>(x + row + 4 * (sub + TPR * k));
      dv[k] = *reinterpret_cast<const f32x4 *>(dy + row + 4 * (sub + TPR * k));
      s += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
    }
    const float mean = group_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[k][j] -= mean;
        q = fmaf(xv[k][j], xv[k][j], q);
      }
    const float rstd = rsqrtf(group_sum(q) * inv_c + eps);
    float g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = xv[k][j] * rstd, d = dv[k][j];
        xv[k][j] = xh;
        dv[k][j] = d * sc[k][j];
        g1 += dv[k][j];
        g2 = fmaf(dv[k][j], xh, g2);
        acs[k][j] = fmaf(d, xh, acs[k][j]);
        act[k][j] += d;
      }
    const float m1 = group_sum(g1) * inv_c, m2 = group_sum(g2) * inv_c;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rstd * (dv[k][j] - m1 - xv[k][j] * m2);
      if (dx_add) o += *reinterpret_cast<const f32x4 *>(dx_add + row + 4 * (sub + TPR * k));
      *reinterpret_cast<f32x4 *>(dx + row + 4 * (sub + TPR * k)) = o;
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * (sub + TPR * k) + j;
      red[grp * C + c] = acs[k][j];
      red[(NG + grp) * C + c] = act[k][j];
    }
  __syncthreads();
  float *o = dss_part + ((size_t)b * nchunk + chunk) * 2 * C;
  for (int c = tid; c < 2 * C; c += 256) {   // row groups in a fixed order
    const float *rp = red + (c < C ? c : NG * C + (c - C));
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) t += rp[g * C];
    o[c] = t;
  }
}

// out[b][j] = sum over chunks of part[b][chunk][j]
// (32 outputs per workgroup, eight threads per output across the chunks: up to 256 chunks per clip)
__global__ __launch_bounds__(256) void chunks_reduce_kernel(const float *__restrict__ part, int nchunk, int cols, float *__restrict__ out) {
  __shared__ float sh[256];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31), b = blockIdx.y;
  const bool valid = col < cols;
  const float v = slice_sum_8(part + (size_t)b * nchunk * cols, nchunk, (size_t)cols, (size_t)(valid ? col : 0), valid, sh);
  if (threadIdx.x < 32 && valid) out[(size_t)b * cols + col] = v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// GroupNorm + SiLU backward, chunked (the per-(clip, group) kernel above has B * G workgroups: 32 at the training batch).  A
// workgroup owns a chunk of rows of one clip over ALL channels (contiguous 16-byte accesses; a thread always meets the same 4
// channels because C / 4 divides 256):
//   gn_stats (norms.hip)   per-chunk (mean, M2) per group
//   gn_bwd_part_kernel     per-chunk sums: dgamma, dbeta per channel; s1 = sum g, s2 = sum g xhat per group   (g = du * gamma)
//   gn_bwd_dx_kernel       dx = rstd (g - mean(g) - xhat mean(g xhat))
// Chunk partials are merged in chunk order by every consumer (Chan for the statistics): no atomics.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gn_group_stats(const float *__restrict__ slab_b, int nch, int G, int chunk_rows, int L, int cpg, float eps,
                                               float *mean_s, float *rstd_s) {
  // (the chunk statistics are fetched eight at a time and merged in chunk order: one memory round trip per eight chunks instead of one per
  //  chunk -- with 16-64 chunks this serial walk was most of the ~10 us floor of the GroupNorm backward launches)
  for (int g = threadIdx.x; g < G; g += 256) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int ch0 = 0; ch0 < nch; ch0 += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = ch0 + u < nch ? *reinterpret_cast<const float2 *>(slab_b + ((size_t)(ch0 + u) * G + g) * 2) : make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ch = ch0 + u;
        if (ch < nch) {
          const float nb = (float)min(chunk_rows, L - ch * chunk_rows) * (float)cpg;
          const float delta = v[u].x - mean, tot = n + nb;
          mean += delta * (nb / tot);
          m2 += v[u].y + delta * delta * (n * nb / tot);
          n = tot;
        }
      }
    }
    mean_s[g] = mean;
    rstd_s[g] = rsqrtf(m2 / n + eps);
  }
}

template <int V>
__global__ __launch_bounds__(256) void gn_bwd_part_kernel(const float *__restrict__ x, const float *__restrict__ da, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ slab, int L, int C, int G, int nch,
                                                          int chunk_rows, float eps, float *__restrict__ dgb_part /* [B][nch][2][C] */,
                                                          float *__restrict__ s12_part /* [B][nch][G][2] */) {
  __shared__ float mean_s[256], rstd_s[256];
  __shared__ float red[4 * V * 256];   // [quantity * V + j][thread]
  __shared__ float chan[4 * 1024];   // [quantity][channel]: dgamma, dbeta, s1, s2 of this chunk
  const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G, vpr = C / V, cv = tid % vpr, rstep = 256 / vpr;
  gn_group_stats(slab + (size_t)b * nch * G * 2, nch, G, chunk_rows, L, cpg, eps, mean_s, rstd_s);
  __syncthreads();
  const int c0 = cv * V;
  float gam[V], bet[V], mu[V], rs[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int g = (c0 + j) / cpg;
    gam[j] = gamma[c0 + j];
    bet[j] = beta[c0 + j];
    mu[j] = mean_s[g];
    rs[j] = rstd_s[g];
  }
  const int r0 = ch * chunk_rows, rows = min(chunk_rows, L - r0);
  const float *xb = x + ((size_t)b * L + r0) * C + c0, *db = da + ((size_t)b * L + r0) * C + c0;
  float a[4][V];
#pragma unroll
  for (int qn = 0; qn < 4; ++qn)
#pragma unroll
    for (int j = 0; j < V; ++j) a[qn][j] = 0.f;
#pragma unroll 4
  for (int r = tid / vpr; r < rows; r += rstep) {
    float xv[V], dv[V];
    if constexpr (V == 4) {
      const f32x4 x4 = *reinterpret_cast<const f32x4 *>(xb + (size_t)r * C), d4 = *reinterpret_cast<const f32x4 *>(db + (size_t)r * C);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[j] = x4[j];
        dv[j] = d4[j];
      }
    } else {
      xv[0] = xb[(size_t)r * C];
      dv[0] = db[(size_t)r * C];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float xh = (xv[j] - mu[j]) * rs[j];
      const float u = fmaf(xh, gam[j], bet[j]);
      const float sg = 1.0f / (1.0f + expf(-u));
      const float du = dv[j] * sg * (1.0f + u * (1.0f - sg));
      const float g1 = du * gam[j];
      a[0][j] = fmaf(du, xh, a[0][j]);
      a[1][j] += du;
      a[2][j] += g1;
      a[3][j] = fmaf(g1, xh, a[3][j]);
    }
  }
#pragma unroll
  for (int qn = 0; qn < 4; ++qn)
#pragma unroll
    for (int j = 0; j < V; ++j) red[(qn * V + j) * 256 + tid] = a[qn][j];
  __syncthreads();
  for (int o = tid; o < 4 * C; o += 256) {   // per-channel totals: the rstep threads that share a column set, in thread order
    const int qn = o / C, c = o - qn * C;
    const float *rp = red + (qn * V + (c % V)) * 256 + (c / V);
    float t = 0.f;
    for (int sl = 0; sl < rstep; ++sl) t += rp[sl * vpr];
    chan[o] = t;
    if (qn < 2) dgb_part[(((size_t)b * nch + ch) * 2 + qn) * C + c] = t;
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      s1 += chan[2 * C + c];
      s2 += chan[3 * C + c];
    }
    s12_part[(((size_t)b * nch + ch) * G + g) * 2] = s1;
    s12_part[(((size_t)b * nch + ch) * G + g) * 2 + 1] = s2;
  }
}

template <int V>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const float *__restrict__ x, const float *__restrict__ da, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, const float *__restrict__ slab,
                                                        const float *__restrict__ s12_part, int L, int C, int G, int nch, int chunk_rows, float eps,
                                                        float *__restrict__ dx, const int B, const TailReduce tail /* dgamma | dbeta slices */,
                                                        const float *__restrict__ dx_add /* or null: a second gradient of x, added here */) {
  __shared__ float mean_s[256], rstd_s[256], m1_s[256], m2_s[256];
  if ((int)blockIdx.y >= B) {   // appended rows of the grid: the slice reduction of the affine gradients (written by gn_bwd_part_kernel)
    const int blk = ((int)blockIdx.y - B) * nch + (int)blockIdx.x;
    if (blk < tail.blocks()) tail_reduce_block(tail, blk, mean_s);
    return;
  }
  const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G, vpr = C / V, cv = tid % vpr, rstep = 256 / vpr;
  gn_group_stats(slab + (size_t)b * nch * G * 2, nch, G, chunk_rows, L, cpg, eps, mean_s, rstd_s);
  for (int g = tid; g < G; g += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int k0 = 0; k0 < nch; k0 += 8) {   // eight chunk sums per memory round trip, added in chunk order
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = k0 + u < nch ? *reinterpret_cast<const float2 *>(s12_part + (((size_t)b * nch + k0 + u) * G + g) * 2) : make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < nch) {
          s1 += v[u].x;
          s2 += v[u].y;
        }
    }
    const float inv_n = 1.0f / ((float)L * (float)cpg);
    m1_s[g] = s1 * inv_n;
    m2_s[g] = s2 * inv_n;
  }
  __syncthreads();
  const int c0 = cv * V;
  float gam[V], bet[V], mu[V], rs[V], m1[V], m2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int g = (c0 + j) / cpg;
    gam[j] = gamma[c0 + j];
    bet[j] = beta[c0 + j];
    mu[j] = mean_s[g];
    rs[j] = rstd_s[g];
    m1[j] = m1_s[g];
    m2[j] = m2_s[g];
  }
  const int r0 = ch * chunk_rows, rows = min(chunk_rows, L - r0);
  const size_t base = ((size_t)b * L + r0) * C + c0;
#pragma unroll 4
  for (int r = tid / vpr; r < rows; r += rstep) {
    float xv[V], dv[V], o[V];
    if constexpr (V == 4) {
      const f32x4 x4 = *reinterpret_cast<const f32x4 *>(x + base + (size_t)r * C), d4 = *reinterpret_cast<const f32x4 *>(da + base + (size_t)r * C);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[j] = x4[j];
        dv[j] = d4[j];
      }
    } else {
      xv[0] = x[base + (size_t)r * C];
      dv[0] = da[base + (size_t)r * C];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float xh = (xv[j] - mu[j]) * rs[j];
      const float u = fmaf(xh, gam[j], bet[j]);
      const float sg = 1.0f / (1.0f + expf(-u));
      const float g1 = dv[j] * sg * (1.0f + u * (1.0f - sg)) * gam[j];
      o[j] = rs[j] * (g1 - m1[j] - xh * m2[j]);
    }
    if (dx_add) {
      if constexpr (V == 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(dx_add + base + (size_t)r * C);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += a4[j];
      } else {
        o[0] += dx_add[base + (size_t)r * C];
      }
    }
    if constexpr (V == 4) *reinterpret_cast<f32x4 *>(dx + base + (size_t)r * C) = f32x4{o[0], o[1], o[2], o[3]};
    else dx[base + (size_t)r * C] = o[0];
  }
}

// a = SiLU(GroupNorm(x)) from the chunk statistics (the activation the weight gradient multiplies), same mapping as above
template <int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       const float *__restrict__ slab, int L, int C, int G, int nch, int chunk_rows, float eps,
                                                       float *__restrict__ out) {
  __shared__ float mean_s[256], rstd_s[256];
  const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G, vpr = C / V, cv = tid % vpr, rstep = 256 / vpr;
  gn_group_stats(slab + (size_t)b * nch * G * 2, nch, G, chunk_rows, L, cpg, eps, mean_s, rstd_s);
  __syncthreads();
  const int c0 = cv * V;
  float sc[V], sh[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int g = (c0 + j) / cpg;
    sc[j] = rstd_s[g] * gamma[c0 + j];
    sh[j] = fmaf(-mean_s[g], sc[j], beta[c0 + j]);
  }
  const int r0 = ch * chunk_rows, rows = min(chunk_rows, L - r0);
  const size_t base = ((size_t)b * L + r0) * C + c0;
#pragma unroll 4
  for (int r = tid / vpr; r < rows; r += rstep) {
    if constexpr (V == 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + base + (size_t)r * C);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = fmaf(xv[j], sc[j], sh[j]);
        o[j] = u / (1.0f + expf(-u));
      }
      *reinterpret_cast<f32x4 *>(out + base + (size_t)r * C) = o;
    } else {
      const float u = fmaf(x[base + (size_t)r * C], sc[0], sh[0]);
      out[base + (size_t)r * C] = u / (1.0f + expf(-u));
    }
  }
}

}  // namespace

hipError_t launch_pack_train(const float *w, int N, int C, int taps, float *fw, void *fwx, float *dg, void *dgx, hipStream_t s) {
  if (taps < 1 || taps > PT_MAX_TAPS) return hipErrorInvalidValue;
  if ((fwx && (((int64_t)taps * C) % 32)) || (dgx && (((int64_t)taps * N) % 32))) return hipErrorInvalidValue;
  if (!fw && !fwx && !dg && !dgx) return hipSuccess;
  const dim3 grid((C + 31) / 32, (N + 31) / 32);
  if (taps == 1) hipLaunchKernelGGL(pack_train_kernel<1>, grid, dim3(256), 0, s, w, N, C, taps, fw, static_cast<f16 *>(fwx), dg, static_cast<bf16 *>(dgx));
  else if (taps == 3) hipLaunchKernelGGL(pack_train_kernel<3>, grid, dim3(256), 0, s, w, N, C, taps, fw, static_cast<f16 *>(fwx), dg, static_cast<bf16 *>(dgx));
  else hipLaunchKernelGGL(pack_train_kernel<0>, grid, dim3(256), 0, s, w, N, C, taps, fw, static_cast<f16 *>(fwx), dg, static_cast<bf16 *>(dgx));
  return hipGetLastError();
}

hipError_t launch_pack_train_many(const void *desc_dev, int n_items, int total_tiles, hipStream_t s) {
  if (!desc_dev || n_items < 1 || total_tiles < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_train_many_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, static_cast<const unsigned long long *>(desc_dev), n_items);
  return hipGetLastError();
}

hipError_t launch_pack_dgrad(const float *w, int N, int C, int taps, int ldn, float *out, hipStream_t s, void *outx) {
  const int64_t total = (int64_t)C * taps * ldn;
  if (outx && (((int64_t)taps * ldn) % 32)) return hipErrorInvalidValue;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid), dim3(256), 0, s, w, N, C, taps, ldn, out, static_cast<bf16 *>(outx));
  return hipGetLastError();
}

// kernel family: 0 = one 32 x 32 tile per workgroup (thin layers), 1 / 2 = LDS-staged 64 x 64 / 128 x 128 tiles
static int wgrad_family(int C, int N, int taps) {
  const int Q = taps * C;
  if (N < 64 || Q < 64 || (C % 32) || (N % 4) || taps > 9) return 0;
  const int TW = (N >= 128 && Q >= 128) ? 2 : 1;
  const int T = 64 * TW;
  if (!(taps == 1 || (C % T) == 0 || C <= T)) return TW == 2 && (C % 64 == 0 || C <= 64) ? 1 : 0;
  return TW;
}

int conv_wgrad_splits(int64_t rows, int C, int N, int taps) {
  const int fam = wgrad_family(C, N, taps), Q = taps * C;
  const int T = fam ? 64 * fam : 32;
  const int64_t tiles = (int64_t)((N + T - 1) / T) * ((Q + T - 1) / T);
  static const int64_t target = [] {   // tuning hook: workgroups per launch the row splits aim at (LDS-staged families)
    const char *e = tune_env("SF_WGRAD_TARGET");
    return e ? (int64_t)atol(e) : 512;
  }();
  int64_t S = std::max<int64_t>(1, (fam ? target : 2048) / std::max<int64_t>(tiles, 1));
  S = std::min<int64_t>(S, std::max<int64_t>(1, rows / 256));
  return (int)std::min<int64_t>(S, 512);   // >= 2 workgroups per CU; more splits only lengthen the reduction
}

hipError_t launch_conv_wgrad(const float *dy, const float *act, int B, int L, int C, int N, int taps, int pad, float *partial, int S, float *dw,
                             hipStream_t s, int x3, const float *bias_part, int bias_slices, float *db, bool *bias_done) {
  if (bias_done) *bias_done = false;
  const int rows = B * L, Q = taps * C;
  int rps = (rows + S - 1) / S;
  rps = (rps + 31) / 32 * 32;
  float *direct = S == 1 ? dw : nullptr;
  const int fam = wgrad_family(C, N, taps);
  // (the gradients dy span the whole fp32 exponent range: the bf16 split, whatever the forward pass uses)
  if (x3 && fam == 2)
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<2, X3_BF16>), dim3((N + 127) / 128, (Q + 127) / 128, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
  else if (x3 && fam == 1)
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<1, X3_BF16>), dim3((N + 63) / 64, (Q + 63) / 64, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
  else if (fam == 2)
    hipLaunchKernelGGL(conv_wgrad_lds_kernel<2>, dim3((N + 127) / 128, (Q + 127) / 128, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
  else if (fam == 1)
    hipLaunchKernelGGL(conv_wgrad_lds_kernel<1>, dim3((N + 63) / 64, (Q + 63) / 64, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
  else {
    // thin layers: one workgroup takes up to three 32-column tiles of (tap, channel) -- the three taps of a 32-channel level: dy is fetched
    // once instead of once per tile, the activation rows of the taps overlap in the cache (same accumulation order per tile: same bits)
    const int qt = (Q + 31) / 32;
    if (qt >= 3) hipLaunchKernelGGL((conv_wgrad_kernel<1, 3>), dim3((N + 31) / 32, (qt + 2) / 3, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
    else if (qt == 2) hipLaunchKernelGGL((conv_wgrad_kernel<1, 2>), dim3((N + 31) / 32, 1, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
    else hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), dim3((N + 31) / 32, 1, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial, direct);
  }
  if (!direct) {
    const int64_t total = (int64_t)N * Q;
    TailReduce tail;   // the bias gradient's slice sums (launch_col_sums_part ran before this call) ride on the reducer
    if (bias_part && db) {
      tail.part = bias_part;
      tail.out = db;
      tail.S = bias_slices;
      tail.cols = N;
      if (bias_done) *bias_done = true;
    }
    // many slices of a small matrix (the thin levels: 512 row splits of 8 x 24 ... 32 x 96 outputs): eight threads per output walk the
    // slices side by side; the vector kernel's thread walks ALL slices of its four outputs, 128 dependent round trips at 512 slices
    // (35-45 us per launch, as much as the weight-gradient kernel itself)
    if (C % 4 == 0 && (S < 64 || total > 65536)) {
      const int nb = (int)((total / 4 + 255) / 256);
      hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3((unsigned)(nb + tail.blocks())), dim3(256), 0, s, partial, S, N, C, taps, dw, nb, tail);
    } else {
      const int nb = (int)((total + 31) / 32);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nb + tail.blocks())), dim3(256), 0, s, partial, S, N, C, taps, dw, nb, tail);
    }
  }
  return hipGetLastError();
}

// out[b][c] = sum_l x[b][l][c] (* y[b][l][c]): the length reductions of the training composition's broadcast adds / SkipModulate scale
// (ATen's strided reduction over the middle dimension runs at ~0.3 TB/s on the wide levels).  Two deterministic stages like col_sums:
// part[b][slice][c] by a (slice, clip) grid, then the slices of a clip in index order.
template <int V, bool MUL>
__global__ __launch_bounds__(256) void length_sums_kernel(const float *__restrict__ x, const float *__restrict__ y, int L, int cols, int rows_per_slice,
                                                          float *__restrict__ part) {
  __shared__ float red[V * 256];
  const int tid = threadIdx.x, vpr = cols / V, cv = tid % vpr, rstep = 256 / vpr;
  const size_t base = (size_t)blockIdx.y * L * cols;
  const int r0 = blockIdx.x * rows_per_slice, r1 = min(L, r0 + rows_per_slice);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int r = r0 + tid / vpr; r < r1; r += rstep) {
    const size_t o = base + (size_t)r * cols + V * cv;
    if constexpr (V == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(x + o);
      if constexpr (MUL) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(y + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], w[j], acc[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
    } else {
      acc[0] = MUL ? fmaf(x[o], y[o], acc[0]) : acc[0] + x[o];
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[j * 256 + tid] = acc[j];
  __syncthreads();
  float *dst = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * cols;
  for (int c = tid; c < cols; c += 256) {
    const float *rp = red + (c % V) * 256 + (c / V);
    float t = 0.f;
    for (int sl = 0; sl < rstep; ++sl) t += rp[sl * vpr];
    dst[c] = t;
  }
}
// any channel count (96, 192, 320, 3, 6 ...): one column per lane, the rows of a slice in order
template <bool MUL>
__global__ __launch_bounds__(64) void length_sums_generic_kernel(const float *__restrict__ x, const float *__restrict__ y, int L, int cols,
                                                                 int rows_per_slice, float *__restrict__ part) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cols) return;
  const size_t base = (size_t)blockIdx.z * L * cols + c;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(L, r0 + rows_per_slice);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) {
    const size_t o = base + (size_t)r * cols;
    acc = MUL ? fmaf(x[o], y[o], acc) : acc + x[o];
  }
  part[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * cols + c] = acc;
}
__global__ __launch_bounds__(256) void length_sums_reduce_kernel(const float *__restrict__ part, int S, int cols, float *__restrict__ out) {
  __shared__ float sh[256];   // 32 outputs per workgroup, 8 threads per output over the slices, combined in a fixed order
  const int col = blockIdx.x * 32 + (threadIdx.x & 31);
  const bool valid = col < cols;
  const float v = slice_sum_8(part + (size_t)blockIdx.y * S * cols, S, (size_t)cols, (size_t)(valid ? col : 0), valid, sh);
  if (threadIdx.x < 32 && valid) out[(size_t)blockIdx.y * cols + col] = v;
}

int length_sums_slices(int B, int L) {   // ~512 workgroups per launch, at least 64 rows each
  const int want = (512 + B - 1) / B;
  return max(1, min(want, (L + 63) / 64));
}
// the vectorised forms; every other channel count runs the one-column-per-lane kernel
static bool length_sums_vec4(int C) { return C % 4 == 0 && C <= 1024 && (256 % (C / 4)) == 0; }
static bool length_sums_vec1(int C) { return C <= 256 && (256 % C) == 0; }

hipError_t launch_length_sums(const float *x, const float *y, int B, int L, int C, float *part, float *out, hipStream_t s) {
  if (B < 1 || L < 1 || C < 1) return hipErrorInvalidValue;
  const int S = length_sums_slices(B, L), rps = (L + S - 1) / S;
  const dim3 grid(S, B);
  if (!length_sums_vec4(C) && !length_sums_vec1(C)) {
    const dim3 gg((C + 63) / 64, S, B);
    if (y) hipLaunchKernelGGL(length_sums_generic_kernel<true>, gg, dim3(64), 0, s, x, y, L, C, rps, part);
    else hipLaunchKernelGGL(length_sums_generic_kernel<false>, gg, dim3(64), 0, s, x, y, L, C, rps, part);
  } else if (length_sums_vec4(C)) {
    if (y) hipLaunchKernelGGL((length_sums_kernel<4, true>), grid, dim3(256), 0, s, x, y, L, C, rps, part);
    else hipLaunchKernelGGL((length_sums_kernel<4, false>), grid, dim3(256), 0, s, x, y, L, C, rps, part);
  } else {
    if (y) hipLaunchKernelGGL((length_sums_kernel<1, true>), grid, dim3(256), 0, s, x, y, L, C, rps, part);
    else hipLaunchKernelGGL((length_sums_kernel<1, false>), grid, dim3(256), 0, s, x, y, L, C, rps, part);
  }
  hipLaunchKernelGGL(length_sums_reduce_kernel, dim3((C + 31) / 32, B), dim3(256), 0, s, part, S, C, out);
  return hipGetLastError();
}

// the two stages of launch_col_sums apart: the slice sums, and their reduction (which the backward pass lets ride on the weight
// gradient's reducer, launch_conv_wgrad)
hipError_t launch_col_sums_part(const float *x, int64_t rows, int cols, float *part, int S, hipStream_t s) {
  const int64_t rps = (rows + S - 1) / S;
  if (cols % 4 == 0 && cols <= 1024 && (256 % (cols / 4)) == 0) hipLaunchKernelGGL(col_sums_vec_kernel<4>, dim3(S), dim3(256), 0, s, x, rows, cols, rps, part);
  else if (cols <= 256 && (256 % cols) == 0) hipLaunchKernelGGL(col_sums_vec_kernel<1>, dim3(S), dim3(256), 0, s, x, rows, cols, rps, part);
  else hipLaunchKernelGGL(col_sums_kernel, dim3((cols + 63) / 64, S), dim3(64), 0, s, x, rows, cols, rps, part);
  return hipGetLastError();
}
hipError_t launch_slices_reduce(const float *part, int S, int cols, float *out, hipStream_t s) {
  hipLaunchKernelGGL(slices_reduce_kernel, dim3((cols + 31) / 32), dim3(256), 0, s, part, S, cols, out);
  return hipGetLastError();
}
hipError_t launch_col_sums(const float *x, int64_t rows, int cols, float *part, int S, float *out, hipStream_t s) {
  hipError_t e = launch_col_sums_part(x, rows, cols, part, S, s);
  return e != hipSuccess ? e : launch_slices_reduce(part, S, cols, out, s);
}

static int gn_bwd_vec(int C, int G) {   // columns per access of the chunked kernels; 0 = unsupported shape
  if (G < 1 || G > 256 || C % G) return 0;
  if (C % 4 == 0 && C <= 1024 && (256 % (C / 4)) == 0) return 4;
  if (C <= 256 && (256 % C) == 0) return 1;
  return 0;
}
static bool gn_bwd_chunked_ok(int C, int G) { return gn_bwd_vec(C, G) != 0; }

static void gn_bwd_plan(int L, int C, int &nch, int &chunk_rows) {
  // (>= 8 K elements per workgroup, <= 64 chunks per clip.  Measured on the training step's four GroupNorm kernels, ms per step: 16 K / 64
  //  6.86, 8 K / 64 6.46, 8 K / 128 6.85, 4 K / 256 10.5 -- every workgroup merges all chunk statistics of its clip in chunk order;
  //  profiles/r6_m_prof_train_gn2.txt.  Before the merges fetched eight chunks per round trip: 7.75 / 8.0 / 9.1, r6_g_prof_train_gn.txt)
  int64_t n = ((int64_t)L * C + 8191) / 8192;
  n = std::max<int64_t>(1, std::min<int64_t>(n, 64));
  chunk_rows = (int)((L + n - 1) / n);
  nch = (L + chunk_rows - 1) / chunk_rows;
}

int64_t gn_silu_bwd_ws_floats(int B, int L, int C, int G) {
  if (!gn_bwd_chunked_ok(C, G)) return (int64_t)B * 2 * C;
  int nch, rows;
  gn_bwd_plan(L, C, nch, rows);
  return (int64_t)B * nch * (2 * C + 4 * G);
}

// Backward of a = SiLU(GroupNorm(x)) in three steps that share the chunk statistics (ws = gn_silu_bwd_ws_floats() floats):
//   launch_gn_silu_recompute: statistics + the activation a (for the weight gradient);  launch_gn_silu_bwd: dx, dgamma, dbeta from da.
hipError_t launch_gn_silu_recompute(const float *x, const float *gamma, const float *beta, int B, int L, int C, int G, float eps, float *act,
                                    float *ws, hipStream_t s) {
  if (G < 1 || C % G) return hipErrorInvalidValue;
  if (!gn_bwd_chunked_ok(C, G)) return launch_gn_silu(F32, x, C, B, L, C, G, gamma, beta, eps, act, C, s);
  int nch, chunk_rows;
  gn_bwd_plan(L, C, nch, chunk_rows);
  hipError_t e = launch_gn_stats(F32, x, C, B, L, C, G, nch, chunk_rows, ws, s);
  if (e != hipSuccess) return e;
  if (gn_bwd_vec(C, G) == 4) hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(nch, B), dim3(256), 0, s, x, gamma, beta, ws, L, C, G, nch, chunk_rows, eps, act);
  else hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(nch, B), dim3(256), 0, s, x, gamma, beta, ws, L, C, G, nch, chunk_rows, eps, act);
  return hipGetLastError();
}

// the statistics launch_gn_silu_bwd reads, alone (the forward pass kept SiLU(GroupNorm(x)) itself); nothing to do where its per-group kernel
// recomputes them
hipError_t launch_gn_bwd_stats(const float *x, int B, int L, int C, int G, float *ws, hipStream_t s) {
  if (G < 1 || C % G) return hipErrorInvalidValue;
  if (!gn_bwd_chunked_ok(C, G)) return hipSuccess;
  int nch, chunk_rows;
  gn_bwd_plan(L, C, nch, chunk_rows);
  return launch_gn_stats(F32, x, C, B, L, C, G, nch, chunk_rows, ws, s);
}

int64_t gn_bwd_stats_floats(int B, int L, int C, int G) {   // size of the chunk statistics alone (0: the per-group kernels keep none)
  if (G < 1 || C % G || !gn_bwd_chunked_ok(C, G)) return 0;
  int nch, rows;
  gn_bwd_plan(L, C, nch, rows);
  return (int64_t)B * nch * G * 2;
}

hipError_t launch_gn_silu_bwd(const float *x, const float *da, const float *gamma, const float *beta, int B, int L, int C, int G, float eps,
                              float *dx, float *ws /* statistics already there (launch_gn_silu_recompute) */, float *dgb /* [2C] = dgamma | dbeta */,
                              hipStream_t s, const float *slab_in /* the statistics kept by the forward pass instead of those in ws */,
                              const float *dx_add /* dx = the GroupNorm gradient + this tensor (a residual branch's gradient of x), or null */) {
  if (G < 1 || C % G) return hipErrorInvalidValue;
  if (gn_bwd_chunked_ok(C, G)) {
    int nch, chunk_rows;
    gn_bwd_plan(L, C, nch, chunk_rows);
    float *dgb_part = ws + (size_t)B * nch * G * 2, *s12 = dgb_part + (size_t)B * nch * 2 * C;
    const float *slab = slab_in ? slab_in : ws;
    TailReduce tail;   // [dgamma | dbeta] = column sums of the B * nch partial rows: rides on the dx launch (rows appended to its grid)
    tail.part = dgb_part;
    tail.out = dgb;
    tail.S = B * nch;
    tail.cols = 2 * C;
    const int extra = (tail.blocks() + nch - 1) / nch;
    if (gn_bwd_vec(C, G) == 4) {
      hipLaunchKernelGGL(gn_bwd_part_kernel<4>, dim3(nch, B), dim3(256), 0, s, x, da, gamma, beta, slab, L, C, G, nch, chunk_rows, eps, dgb_part, s12);
      hipLaunchKernelGGL(gn_bwd_dx_kernel<4>, dim3(nch, B + extra), dim3(256), 0, s, x, da, gamma, beta, slab, s12, L, C, G, nch, chunk_rows, eps, dx, B, tail, dx_add);
    } else {
      hipLaunchKernelGGL(gn_bwd_part_kernel<1>, dim3(nch, B), dim3(256), 0, s, x, da, gamma, beta, slab, L, C, G, nch, chunk_rows, eps, dgb_part, s12);
      hipLaunchKernelGGL(gn_bwd_dx_kernel<1>, dim3(nch, B + extra), dim3(256), 0, s, x, da, gamma, beta, slab, s12, L, C, G, nch, chunk_rows, eps, dx, B, tail, dx_add);
    }
    return hipGetLastError();
  }
  const int cpg = C / G;
  if (cpg > 256 || (256 % cpg)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_silu_bwd_kernel, dim3(B * G), dim3(256), 0, s, x, da, gamma, beta, L, C, G, eps, dx, ws, dx_add);
  // ws is [B][2][C]: rows b, columns (2C) -> column sums give [dgamma | dbeta]
  hipLaunchKernelGGL(slices_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, s, ws, B, 2 * C, dgb);
  return hipGetLastError();
}

// >= 8 K elements per workgroup (at least 8 rows), at most 256 chunks per clip.  (64-row chunks before: the 1024-channel levels of the
// training step ran 4 clips x 4 chunks = 16 workgroups, 40 us for 16 MB of traffic.)
int ln_mod_bwd_chunks(int L, int C) {
  const int rows = std::max(8, 8192 / std::max(C, 1));
  return std::max(1, std::min(256, (L + rows - 1) / rows));
}

hipError_t launch_ln_modulate_bwd(const float *x, const float *ss, const float *dy, float eps, int B, int L, int C, float *dx, float *dss_part,
                                  float *dss, hipStream_t s, const float *dx_add) {
  const int nchunk = ln_mod_bwd_chunks(L, C);
  const int rpc = (L + nchunk - 1) / nchunk;
#define SF_LNB(TPR, PER) hipLaunchKernelGGL((ln_mod_bwd_vec_kernel<TPR, PER>), dim3(B, nchunk), dim3(256), 0, s, x, ss, 2 * C, dy, L, eps, rpc, dx, dss_part, dx_add)
  switch (C) {
    case 4: SF_LNB(1, 1); break;
    case 8: SF_LNB(2, 1); break;
    case 16: SF_LNB(4, 1); break;
    case 32: SF_LNB(8, 1); break;
    case 64: SF_LNB(16, 1); break;
    case 128: SF_LNB(32, 1); break;
    case 256: SF_LNB(64, 1); break;
    case 512: SF_LNB(64, 2); break;
    case 1024: SF_LNB(64, 4); break;
    default: return hipErrorInvalidValue;
  }
#undef SF_LNB
  if (dss) hipLaunchKernelGGL(chunks_reduce_kernel, dim3((2 * C + 31) / 32, B), dim3(256), 0, s, dss_part, nchunk, 2 * C, dss);
  return hipGetLastError();
}

}  // namespace sf
