// Frame-walk (1,3,3) convolution for the first stage of the VideoOnsetNet (R(2+1)D layer 1: 64 input channels, 56 x 56 frames;
// main/resnet.py:43-52 Conv2Plus1D first half, main/onset_net.py:19-36).
//
// As an implicit GEMM the spatial convolution fetches every input row nine times (once per tap) and its 128x192 macro tile computes 192
// columns for the 144 that exist: 748 us per launch at N = 32 clips, matrix cores busy 29 % (profiles/r4_z_pmc_mfma_onset_by_kernel.csv).
// Here a workgroup owns ONE 32-column tile of the output and a PH x PW patch of positions, and walks the frames of a clip:
//   * its 32 x 576 weight slice is register-stationary in MFMA fragment order for the whole walk (36 fragments per wave, every wave of
//     the workgroup holds the same slice: registers are what a CU has most of) -- a K step costs ONE LDS fragment read and one MFMA;
//   * the (PH + 2) x (PW + 2) halo tile of a frame is staged once in LDS and serves all nine taps (a tap is a constant LDS offset);
//     the next frame's halo is in flight in registers while this one is multiplied;
//   * the five column tiles of a patch carry block indices 8 apart, i.e. they run on the same XCD at about the same time: its L2
//     fetches the patch's input once.
//   in : rows ((n T + t) H + h) W + w  x  in_ld (64 channels read), 16-bit;   out: same rows x out_ld, columns [0, 32 * ntiles) written:
//   relu(conv + shift) for columns < n_real, zeros above (the temporal walk reads 160-channel rows).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int PH = 8, PW = 14;                    // patch of output positions per workgroup: 112 (56 x 56 frames tile exactly)
constexpr int HP = (PH + 2) * (PW + 2);           // halo positions: 160
constexpr int CIN = 64, PITCH = CIN + 8;          // 144-byte rows: conflict-free 16-byte fragment reads
constexpr int NVH = HP * (CIN / 8) / 256;         // staging vectors per thread and frame: 5
constexpr int OP = 32 + 8;                        // output staging pitch (elements): 80-byte rows

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_sp_kernel(const T *__restrict__ in, const int in_ld, const T *__restrict__ wfr, const float *__restrict__ shift,
                                                      const int n_real, T *__restrict__ out, const int out_ld, const int Tn, const int H, const int W,
                                                      const int ntiles, const int tiles_w, const int npatch, const int relu) {
  using frag = typename Frag16<T>::type;
  __shared__ __attribute__((aligned(16))) T halo[2][HP * PITCH];
  __shared__ __attribute__((aligned(16))) float shift_s[32];   // this column tile's shifts (zeros above n_real): 16 registers less per lane
  // Output staging: a lane's results are 8-byte pieces of 32 different rows (230 us of a 764 us launch went into those stores); through
  // LDS four lanes write one row's whole 64-byte segment of this column tile.  Two slots: frame t is stored while frame t + 1 is multiplied.
  __shared__ __attribute__((aligned(16))) T ostage[2][128 * OP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int mtile = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;
  // block -> (patch, column tile): groups of 8 * ntiles blocks cover 8 patches; inside a group the block's XCD (index % 8) picks the
  // patch and index / 8 the column tile, so that the column tiles of a patch share an XCD
  const int grp = (int)blockIdx.x / (8 * ntiles), within = (int)blockIdx.x - grp * 8 * ntiles;
  const int patch = grp * 8 + (within & 7), nt = within >> 3;
  if (patch >= npatch) return;
  const int n = blockIdx.y;
  const int h0 = (patch / tiles_w) * PH, w0 = (patch % tiles_w) * PW;
  const size_t frame_rows = (size_t)H * W;
  const size_t clip_row0 = (size_t)n * Tn * frame_rows;

  // ---- weights: 36 register-stationary fragments [nt][tap 9][kc 4][lane][8] ------------------------------------------------------------
  frag wf[36];
  {
    const frag *wp = reinterpret_cast<const frag *>(wfr) + (size_t)nt * 36 * 64 + lane;
#pragma unroll
    for (int s = 0; s < 36; ++s) wf[s] = wp[s * 64];
  }
  // ---- halo staging: thread -> (halo position, 8-channel vector); out-of-image positions read zeros (the convolution's padding) --------
  const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in + clip_row0 * in_ld), 0,
                                                                        (unsigned)((size_t)Tn * frame_rows * in_ld * sizeof(T)), 0x00020000);
  unsigned soff[NVH];
#pragma unroll
  for (int i = 0; i < NVH; ++i) {
    const int idx = tid + 256 * i, hp = idx >> 3, v = idx & 7;
    const int hy = hp / (PW + 2), hx = hp - hy * (PW + 2);
    const int h = h0 - 1 + hy, w = w0 - 1 + hx;
    soff[i] = (h >= 0 && h < H && w >= 0 && w < W) ? (unsigned)(((h * W + w) * in_ld + v * 8) * sizeof(T)) : 0x80000000u;
  }
  const unsigned frame_b = (unsigned)(frame_rows * in_ld * sizeof(T));
  struct Stage {
    Vec16<T> v[NVH];
  };
  auto fetch = [&](Stage &st, int f) {
    const unsigned fo = f < Tn ? (unsigned)f * frame_b : 0x80000000u;
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
      u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rIn, soff[i] | (fo & 0x80000000u), fo & 0x7FFFFFFFu, 0);
      st.v[i].v = __builtin_bit_cast(decltype(st.v[i].v), raw);
    }
  };
  auto stash = [&](const Stage &st, int slot) {
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
      const int idx = tid + 256 * i, hp = idx >> 3, v = idx & 7;
      st16<T>(&halo[slot][hp * PITCH + v * 8], st.v[i]);
    }
  };
  // ---- this lane's output position (MFMA B-operand row) and its epilogue operands --------------------------------------------------------
  const int m = mtile * 32 + fr;                     // 0 .. 127, positions >= PH * PW are dummies
  const int py = min(m / PW, PH - 1), px = min(m - (m / PW) * PW, PW - 1);   // (dummy rows stay inside the halo tile)
  const int hpos = py * (PW + 2) + px;               // halo index of tap (0, 0); tap (dy, dx) adds dy * (PW + 2) + dx
  if (tid < 32) shift_s[tid] = (nt * 32 + tid) < n_real ? shift[nt * 32 + tid] : 0.f;
  Stage s0, s1;
  fetch(s0, 0);
  fetch(s1, 1);
  stash(s0, 0);
  fetch(s0, 2);
  __syncthreads();

  // frame f's staged results -> global: thread -> (row = tid / 4 + 64 j, 16-byte quarter tid % 4) of the 128 x 32 tile
  int orow[2];
  bool ovalid[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int mm = (tid >> 2) + 64 * j;
    const int qy = min(mm / PW, PH - 1), qx = min(mm - (mm / PW) * PW, PW - 1);
    ovalid[j] = mm < PH * PW && (h0 + qy) < H && (w0 + qx) < W;
    orow[j] = (h0 + qy) * W + (w0 + qx);
  }
  auto flush = [&](int f) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int mm = (tid >> 2) + 64 * j;
      const Vec16<T> v = ld16<T>(&ostage[f & 1][mm * OP + (tid & 3) * 8]);
      if (ovalid[j]) st16<T>(out + (clip_row0 + (size_t)f * frame_rows + orow[j]) * out_ld + nt * 32 + (tid & 3) * 8, v);
    }
  };
  auto step = [&](int t, Stage &next, Stage &after) {   // frame t sits in slot t & 1; `next` holds frame t + 1, `after` goes in flight with t + 3
    if (t > 0) flush(t - 1);                             // the previous frame's tile, complete since the barrier that ended its step
    const T *src = &halo[t & 1][hpos * PITCH + fh * 8];
    f32x16 acc, acc2;   // two accumulators: consecutive MFMAs do not wait for each other's result
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kc = 0; kc < 4; kc += 2) {
          const frag af = *reinterpret_cast<const frag *>(src + (dy * (PW + 2) + dx) * PITCH + kc * 16);
          const frag ag = *reinterpret_cast<const frag *>(src + (dy * (PW + 2) + dx) * PITCH + kc * 16 + 16);
          acc = mfma32x16(wf[(dy * 3 + dx) * 4 + kc], af, acc);   // D^T[n][m]: four consecutive channels of one position per lane and group
          acc2 = mfma32x16(wf[(dy * 3 + dx) * 4 + kc + 1], ag, acc2);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    stash(next, (t + 1) & 1);          // the other slot: nobody reads it during this step (frame t - 1 was released by the last barrier)
    {
      T *sp = &ostage[t & 1][m * OP + 4 * fh];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        T o[4];
        const f32x4 bi = *reinterpret_cast<const f32x4 *>(shift_s + 8 * g + 4 * fh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[4 * g + e] + bi[e];
          o[e] = from_f<T>((relu & 1) ? fmaxf(v, 0.f) : v);
        }
        __builtin_memcpy(__builtin_assume_aligned(sp + 8 * g, 8), o, 8);
      }
    }
    fetch(after, t + 3);
    __syncthreads();
  };
  // register sets rotate with period 2: at step t the set holding frame t + 1 is stashed, the OTHER set already holds frame t + 2
  for (int t = 0; t < Tn; t += 2) {
    step(t, s1, s1);        // s1 = frame t + 1 -> slot; then s1 refilled with frame t + 3 (s0 keeps frame t + 2)
    if (t + 1 < Tn) step(t + 1, s0, s0);   // s0 = frame t + 2 -> slot; refilled with frame t + 4
  }
  flush(Tn - 1);
}

// packed [n_real][9 * 64] (compute type, BatchNorm folded) -> fragments [ntiles][tap 9][kc 4][lane 64][8]; rows >= n_real are zeros
template <typename T> __global__ void pack_sp_kernel(const T *__restrict__ w, int n_real, int ntiles, T *__restrict__ out) {
  const int total = ntiles * 36 * 64 * 8;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int q = e & 7, lane = (e >> 3) & 63;
    int t = e >> 9;
    const int kc = t & 3;
    t >>= 2;
    const int tap = t % 9, nt = t / 9;
    const int n = nt * 32 + (lane & 31);
    out[e] = n < n_real ? w[(size_t)n * (9 * CIN) + tap * CIN + kc * 16 + (lane >> 5) * 8 + q] : from_f<T>(0.f);
  }
}

}  // namespace

// applicable: 16-bit type, (1,3,3) kernel, stride 1, padding 1, 64 input channels in 64-channel rows, output rows of >= 32 * ceil(N / 32) channels
bool conv_sp_ok(int dt, int cin_real, int cin_ld, int cout, int out_ld) {
  static const bool off = tune_env("SF_NO_SP") != nullptr;   // A/B aid
  if (off || dt == F32 || cin_real != CIN || cin_ld != CIN || cout < 1) return false;
  const int ntiles = (cout + 31) / 32;
  return out_ld == 32 * ntiles && ntiles <= 8;   // every column of the output rows is written
}
size_t conv_sp_weight_elems(int cout) { return (size_t)((cout + 31) / 32) * 36 * 64 * 8; }

hipError_t launch_pack_conv_sp(int dt, const void *w, int cout, void *out, hipStream_t s) {
  if (dt == F32) return hipErrorInvalidValue;
  const int ntiles = (cout + 31) / 32;
  if (dt == BF16) hipLaunchKernelGGL((pack_sp_kernel<bf16>), dim3(64), dim3(256), 0, s, static_cast<const bf16 *>(w), cout, ntiles, static_cast<bf16 *>(out));
  else hipLaunchKernelGGL((pack_sp_kernel<f16>), dim3(64), dim3(256), 0, s, static_cast<const f16 *>(w), cout, ntiles, static_cast<f16 *>(out));
  return hipGetLastError();
}

hipError_t launch_conv_sp(int dt, const void *in, int in_ld, const void *wfr, const float *shift, int cout, void *out, int out_ld, int N, int T, int H,
                          int W, int relu, hipStream_t s) {
  if (dt == F32 || N < 1 || T < 1 || H < 1 || W < 1) return hipErrorInvalidValue;
  if ((size_t)T * H * W * in_ld * 2 >= 0x7FFFFFF0ull) return hipErrorInvalidValue;
  const int ntiles = (cout + 31) / 32;
  const int tiles_h = (H + PH - 1) / PH, tiles_w = (W + PW - 1) / PW, npatch = tiles_h * tiles_w;
  const dim3 grid(((npatch + 7) / 8) * 8 * ntiles, N);
  if (dt == BF16)
    hipLaunchKernelGGL((conv_sp_kernel<bf16>), grid, dim3(256), 0, s, static_cast<const bf16 *>(in), in_ld, static_cast<const bf16 *>(wfr), shift, cout,
                       static_cast<bf16 *>(out), out_ld, T, H, W, ntiles, tiles_w, npatch, relu);
  else
    hipLaunchKernelGGL((conv_sp_kernel<f16>), grid, dim3(256), 0, s, static_cast<const f16 *>(in), in_ld, static_cast<const f16 *>(wfr), shift, cout,
                       static_cast<f16 *>(out), out_ld, T, H, W, ntiles, tiles_w, npatch, relu);
  return hipGetLastError();
}

}  // namespace sf
