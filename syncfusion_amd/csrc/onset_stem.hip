// VideoOnsetNet stem, spatial half: Conv3d(3 -> 45, kernel (1,7,7), stride (1,2,2), padding (0,3,3)) + folded BatchNorm + ReLU
// (main/resnet.py:177-187 `R2Plus1dStem`) on channels-last frames with the 3 colour planes padded to 4 (8 bytes per pixel).
//
// The generic implicit-GEMM kernels gather this layer's A operand 4 channels at a time (K = 49 taps x 4 = 196) and ran it at
// 19 TFLOP/s -- 8.7 % of the network's forward time for 0.4 % of its FLOPs.  Here the geometry does the gathering:
//   * for a fixed kernel row dh the 7 taps along w are 7 CONSECUTIVE pixels of one input row = 56 contiguous bytes; with an 8th,
//     zero-weighted tap that is 64 bytes = 32 k-values, i.e. two 16-deep MFMA k-steps whose B fragments are plain 16-byte loads
//     (no LDS staging; neighbouring output pixels read overlapping windows, so the loads hit L1 / L2);
//   * the product is computed transposed, D^T = W . A^T (weights as the A operand from LDS, pixels as the B operand), so that
//     a lane owns one output pixel and 4 consecutive channels per accumulator group: 8-byte stores;
//   * K = 7 rows x 32 = 224: 14 k-steps x 2 channel tiles = 28 MFMAs per 32 output pixels.
// 16-bit types only (the fp32 parity path keeps the generic kernel).
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int KH = 7, KW = 7, KWP = 8, CP = 4;     // taps, taps per row incl. the zero tap, padded colour planes
constexpr int KROW = KWP * CP;                      // 32 k-values per kernel row
constexpr int KTOT = KH * KROW;                     // 224
constexpr int NOUT = 64;                            // output channels incl. padding (45 real)
constexpr int WLD = KTOT + 8;                       // LDS row pitch of the weight image (464 B: conflict-free 16-byte reads)

// generic packed stem weights [n][tap = dh*7 + dw][4] (row length k_in) -> [64][dh][8][4], zero for dw == 7 and n >= n_real
template <typename T> __global__ void stem_repack_kernel(const T *__restrict__ w, int n_real, int k_in, T *__restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NOUT * KTOT) return;
  const int n = idx / KTOT, k = idx - n * KTOT;
  const int dh = k / KROW, r = k - dh * KROW, dw = r / CP, c = r - dw * CP;
  out[idx] = (n < n_real && dw < KW) ? w[(size_t)n * k_in + (dh * KW + dw) * CP + c] : from_f<T>(0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void onset_stem_kernel(const T *__restrict__ in, int NT, int H, int W, int Ho, int Wo, const T *__restrict__ wk,
                                                         const float *__restrict__ shift, int n_real, T *__restrict__ out, int out_ld,
                                                         unsigned bytes_in) {
  using frag = typename Frag16<T>::type;
  __shared__ __attribute__((aligned(16))) T wl[NOUT * WLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  for (int i = tid; i < NOUT * (KTOT / 8); i += 256) {   // 16-byte copies of the weight image
    const int n = i / (KTOT / 8), v = i - n * (KTOT / 8);
    st16<T>(wl + n * WLD + v * 8, ld16<T>(wk + (size_t)n * KTOT + v * 8));
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in), 0, bytes_in, 0x00020000);
  // per-lane bias of the channels it will own: tile i, group v, element e -> channel 32 i + 8 v + 4 fh + e
  float bs[2][4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 32 * i + 8 * v + 4 * fh + e;
        bs[i][v][e] = c < n_real ? shift[c] : 0.f;
      }
  const long total = (long)NT * Ho * Wo;
  const long ntile = (total + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntile; tile += (long)gridDim.x * 4) {
    const long m = tile * 32 + fr;
    const bool mv = m < total;
    const long mm = mv ? m : 0;
    const int wo = (int)(mm % Wo);
    const long r1 = mm / Wo;
    const int ho = (int)(r1 % Ho);
    const long nt = r1 / Ho;
    const int wi0 = 2 * wo - 3, hi0 = 2 * ho - 3;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // this lane's two pixels of k-step (dh, s2): wi0 + 4 s2 + 2 fh and the next one
    frag bfr[KH][2];
#pragma unroll
    for (int dh = 0; dh < KH; ++dh) {
      const int hi = hi0 + dh;
      const bool hv = mv && hi >= 0 && hi < H;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int wi = wi0 + 4 * s2 + 2 * fh;
        const bool v0 = hv && wi >= 0 && wi < W, v1 = hv && wi + 1 >= 0 && wi + 1 < W;
        // left border (first pixel outside the row): load from the second pixel and move it into place, so that the load never starts
        // before the buffer; right border: the second half is the next row's first pixel (or past the buffer's end, which the
        // range-checked load returns as zeros) and is cleared
        const long pix = (nt * H + (hv ? hi : 0)) * (long)W + (v0 ? wi : wi + 1);
        const unsigned off = (v0 || v1) ? (unsigned)(pix * (CP * 2)) : 0x80000000u;
        u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rI, off, 0, 0);
        if (!v0) {
          raw[2] = raw[0];
          raw[3] = raw[1];
          raw[0] = 0u;
          raw[1] = 0u;
        }
        if (!v1) {
          raw[2] = 0u;
          raw[3] = 0u;
        }
        bfr[dh][s2] = __builtin_bit_cast(frag, raw);
      }
    }
#pragma unroll
    for (int dh = 0; dh < KH; ++dh)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int k0 = dh * KROW + 16 * s2 + 8 * fh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const frag af = *reinterpret_cast<const frag *>(wl + (32 * i + fr) * WLD + k0);
          acc[i] = mfma32x16(af, bfr[dh][s2], acc[i]);
        }
      }
    // D^T[n][pixel]: lane -> pixel fr (column), registers -> channels 32 i + (r & 3) + 8 (r >> 2) + 4 fh
    if (mv) {
      T *op = out + (size_t)m * out_ld;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          T o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = from_f<T>(fmaxf(acc[i][4 * v + e] + bs[i][v][e], 0.f));
          *reinterpret_cast<uint2 *>(op + 32 * i + 8 * v + 4 * fh) = *reinterpret_cast<const uint2 *>(o);
        }
    }
  }
}

}  // namespace

size_t onset_stem_weight_elems() { return (size_t)NOUT * KTOT; }

hipError_t launch_onset_stem_repack(int dt, const void *w_generic, int n_real, int k_in, void *out, hipStream_t s) {
  if (dt == F32 || n_real > NOUT) return hipErrorInvalidValue;
  const int total = NOUT * KTOT;
  if (dt == F16) hipLaunchKernelGGL((stem_repack_kernel<f16>), dim3((total + 255) / 256), dim3(256), 0, s, (const f16 *)w_generic, n_real, k_in, (f16 *)out);
  else hipLaunchKernelGGL((stem_repack_kernel<bf16>), dim3((total + 255) / 256), dim3(256), 0, s, (const bf16 *)w_generic, n_real, k_in, (bf16 *)out);
  return hipGetLastError();
}

// in: (NT, H, W, 4) channels-last frames; out: (NT * Ho * Wo, out_ld >= 64) with ReLU(conv + shift); pad channels [n_real, 64) = 0
hipError_t launch_onset_stem(int dt, const void *in, int NT, int H, int W, const void *wk, const float *shift, int n_real, void *out, int out_ld,
                             hipStream_t s) {
  if (dt == F32 || out_ld < NOUT || (out_ld % 4)) return hipErrorInvalidValue;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const size_t bytes = (size_t)NT * H * W * CP * 2;
  if (bytes >= 0x7FFFFFF0ull) return hipErrorInvalidValue;
  const long ntile = ((long)NT * Ho * Wo + 31) / 32;
  const int grid = (int)std::min<long>((ntile + 3) / 4, 4096);
  if (dt == F16)
    hipLaunchKernelGGL((onset_stem_kernel<f16>), dim3(grid), dim3(256), 0, s, (const f16 *)in, NT, H, W, Ho, Wo, (const f16 *)wk, shift, n_real, (f16 *)out,
                       out_ld, (unsigned)bytes);
  else
    hipLaunchKernelGGL((onset_stem_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const bf16 *)in, NT, H, W, Ho, Wo, (const bf16 *)wk, shift, n_real,
                       (bf16 *)out, out_ld, (unsigned)bytes);
  return hipGetLastError();
}

}  // namespace sf
