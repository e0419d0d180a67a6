// Temporal-walk (3,1,1) convolution for the wide-spatial levels of the VideoOnsetNet (R(2+1)D stem and layer 1: 56 x 56 frames, 64 output
// channels; main/resnet.py:43-52 Conv2Plus1D second half, main/onset_net.py:19-36 temporal stride removed).
//
// As an implicit GEMM the temporal convolution reads every mid-tensor row three times (once per tap: the taps are whole FRAMES apart, so a
// row tile shares nothing with its neighbours' taps): 3.5 GB through the L2 -> LDS path per launch at N = 32 clips, 7.9 TB/s, matrix cores
// busy 17 % (profiles/r4_h_pmc_mfma_onset_by_kernel.csv, the 256x64 macro tile).  Here a workgroup owns 128 spatial positions of one clip
// and WALKS the frames: three frame tiles live in an LDS ring, each is fetched once and serves three output frames; the 64 x (3 x C)
// weights are register-stationary in MFMA fragment order for the whole walk (30 or 12 fragments per wave); per frame step a wave does one
// 32x32 output tile (30 / 12 MFMAs from LDS fragments), adds bias (+ residual), applies ReLU and stores 8 bytes per lane and row group.
//   in : rows ((n T + t) HW + p) x in_ld, 16-bit; the first CK = 16 * KS channels are read (layer 1: 160 of the 192-padded 144, stem: 64)
//   out: same rows x out_ld (>= 64), 64 channels written
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int TP = 128;   // positions per workgroup

// KS = 16-channel K steps per tap.  512 threads: wave w -> row tile (w & 3) of the 128 positions, column tile (w >> 2) of the 64 outputs.
template <typename T, int KS>
__global__ __launch_bounds__(512) void conv_tw_kernel(const T *__restrict__ in, const int in_ld, const T *__restrict__ wfr, const float *__restrict__ bias,
                                                      const T *__restrict__ res, const int res_ld, T *__restrict__ out, const int out_ld, const int Tn,
                                                      const int HW, const int relu, const int tp) {
  using frag = typename Frag16<T>::type;
  constexpr int CK = 16 * KS, PITCH = CK + 8;       // elements; 336 B / 144 B rows: conflict-free 16-byte fragment reads
  constexpr int VPR = CK / 8;                        // 16-byte vectors per staged row
  constexpr int NV = (TP * VPR + 511) / 512;         // staging vectors per thread and frame
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T *ring = reinterpret_cast<T *>(smem);             // [3][TP][PITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mtile = wave & 3, ntile = wave >> 2;
  const int fr = lane & 31, fh = lane >> 5;
  const int n = blockIdx.y, p0 = blockIdx.x * tp;   // tp <= TP positions per workgroup: chosen so that the grid is whole rounds of 256 CUs
  const int np = min(tp, HW - p0);                   // valid positions of this tile
  const size_t frame_rows = (size_t)HW;
  const size_t row0 = (size_t)n * Tn * frame_rows + p0;   // row of (frame 0, first position)

  // ---- weights: register-stationary fragments [ntile][tap][ks][lane][8] ------------------------------------------------------------
  frag wf[3 * KS];
  {
    const frag *wp = reinterpret_cast<const frag *>(wfr) + (size_t)ntile * 3 * KS * 64 + lane;
#pragma unroll
    for (int s = 0; s < 3 * KS; ++s) wf[s] = wp[s * 64];
  }
  // bias of the 64 output channels: in LDS behind the ring (16 registers less per lane: the walk holds 120 weight registers and two
  // frames in flight inside a 256-register budget)
  float *bias_s = reinterpret_cast<float *>(smem + (size_t)3 * TP * PITCH * sizeof(T));
  if (tid < 64) bias_s[tid] = bias[tid];

  // ---- staging: thread -> (row, vector) pairs of a frame tile.  One workgroup per CU (the ring takes 129 KB), so the walk itself has to
  //      cover the memory latency: TWO frames are in flight in registers (frames t + 2 and t + 3 while frame t is multiplied), and the
  //      residual rows of the next two steps with them.
  struct Stage {
    Vec16<T> v[NV];
  };
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  struct ResRow {
    u32x2 v[4];
  };
  // buffer loads relative to this clip tile's first row: 32-bit offsets, out-of-range (frames outside the clip, positions beyond the
  // tile) return zeros without a branch
  const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in + row0 * in_ld), 0,
                                                                        (unsigned)(((size_t)(Tn - 1) * frame_rows + np) * in_ld * sizeof(T)), 0x00020000);
  unsigned soff[NV];   // (row, vector) byte offset inside a frame tile, or out of range
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + 512 * i, r = idx / VPR, v = idx - r * VPR;
    soff[i] = (idx < TP * VPR && r < np) ? (unsigned)((r * in_ld + v * 8) * sizeof(T)) : 0x80000000u;
  }
  const unsigned frame_b = (unsigned)(frame_rows * in_ld * sizeof(T));
  auto fetch = [&](Stage &st, int f) {   // frame f of this clip -> registers (zeros outside the clip's frames / the tile's valid positions)
    const unsigned fo = (f >= 0 && f < Tn) ? (unsigned)f * frame_b : 0x80000000u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rIn, soff[i] | (fo & 0x80000000u), fo & 0x7FFFFFFFu, 0);
      st.v[i].v = __builtin_bit_cast(decltype(st.v[i].v), raw);
    }
  };
  auto stash = [&](const Stage &st, int slot) {
    T *dst = ring + (size_t)slot * TP * PITCH;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + 512 * i, r = idx / VPR, v = idx - r * VPR;
      if (idx < TP * VPR) st16<T>(dst + r * PITCH + v * 8, st.v[i]);
    }
  };
  const int m = mtile * 32 + fr;                     // this lane's position inside the tile (MFMA B operand row)
  const bool mvalid = m < np;
  // residual rows the same way (a null residual is a resource of zero records: zeros, no branch)
  const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T *>(res ? res + row0 * res_ld : in), 0, res ? (unsigned)(((size_t)(Tn - 1) * frame_rows + np) * res_ld * sizeof(T)) : 0u, 0x00020000);
  const unsigned roff = mvalid ? (unsigned)((m * res_ld + ntile * 32 + 4 * fh) * sizeof(T)) : 0x80000000u;
  const unsigned rframe_b = (unsigned)(frame_rows * res_ld * sizeof(T));
  auto load_res = [&](ResRow &rr, int f) {           // residual of this lane's 16 outputs of frame f
    const unsigned fo = f < Tn ? (unsigned)f * rframe_b : 0x80000000u;
#pragma unroll
    for (int g = 0; g < 4; ++g) rr.v[g] = __builtin_amdgcn_raw_buffer_load_b64(rRes, (roff + (unsigned)(8 * g * sizeof(T))) | (fo & 0x80000000u), fo & 0x7FFFFFFFu, 0);
  };
  // frame f lives in slot (f + 1) % 3: frame -1 (zero padding) in slot 0, frame 0 in slot 1, frame 1 in slot 2
  Stage s0, s1;
  ResRow r0, r1;
  fetch(s0, -1);
  stash(s0, 0);
  fetch(s0, 0);
  fetch(s1, 1);
  stash(s0, 1);
  stash(s1, 2);
  fetch(s0, 2);
  fetch(s1, 3);
  load_res(r0, 0);
  load_res(r1, 1);
  __syncthreads();

  auto step = [&](int t, Stage &st, ResRow &rr) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const T *src = ring + (size_t)((t + tap) % 3) * TP * PITCH + m * PITCH + fh * 8;   // frame t - 1 + tap -> slot (t + tap) % 3
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag af = *reinterpret_cast<const frag *>(src + ks * 16);
        acc = mfma32x16(wf[tap * KS + ks], af, acc);   // D^T[n][m]: a lane ends up with four consecutive channels of one position
      }
    }
    // ---- epilogue of frame t: bias (+ residual), ReLU, 8-byte stores --------------------------------------------------------------
    if (mvalid) {
      const size_t row = row0 + (size_t)t * frame_rows + m;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = ntile * 32 + 8 * g + 4 * fh;
        T rv[4];
        __builtin_memcpy(rv, &rr.v[g], 8);
        float v[4];
        const f32x4 bi = *reinterpret_cast<const f32x4 *>(bias_s + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e] + bi[e] + to_f(rv[e]);
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f<T>(relu ? fmaxf(v[e], 0.f) : v[e]);
        __builtin_memcpy(__builtin_assume_aligned(out + row * out_ld + c, 8), o, 8);
      }
    }
    load_res(rr, t + 2);           // this register set serves step t + 2 next
    __syncthreads();               // every wave is done with slot (t % 3) = frame t - 1
    stash(st, t % 3);              // frame t + 2 takes it ((t + 3) % 3)
    fetch(st, t + 4);              // ... and the set goes back in flight with frame t + 4
    __syncthreads();
  };
  for (int t = 0; t < Tn; t += 2) {
    step(t, s0, r0);
    if (t + 1 < Tn) step(t + 1, s1, r1);
  }
}

// [N = 64][3 taps x cin_ld] (compute type) -> fragments [ntile 2][tap 3][ks KS][lane 64][8]: lane -> W[32 ntile + lane % 32][tap][16 ks + 8 (lane / 32) + 0..7]
template <typename T> __global__ void pack_tw_kernel(const T *__restrict__ w, int cin_ld, int KS, T *__restrict__ out) {
  const int total = 2 * 3 * KS * 64 * 8;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int q = e & 7, lane = (e >> 3) & 63;
    int t = e >> 9;
    const int ks = t % KS;
    t /= KS;
    const int tap = t % 3, ntile = t / 3;
    out[e] = w[(size_t)(ntile * 32 + (lane & 31)) * (3 * cin_ld) + tap * cin_ld + ks * 16 + (lane >> 5) * 8 + q];
  }
}

int tw_ksteps(int cin_real) { return cin_real <= 64 ? 4 : (cin_real <= 160 ? 10 : 0); }

}  // namespace

// applicable: 16-bit type, 64 output channels, (3,1,1) kernel with temporal padding 1, channel counts the two instantiations cover
bool conv_tw_ok(int dt, int cin_real, int cin_ld, int cout, int out_ld, int res_ld) {
  static const bool off = tune_env("SF_NO_TW") != nullptr;   // A/B aid
  if (off || dt == F32 || cout != 64 || out_ld < 64 || (out_ld % 4) || (res_ld % 4)) return false;
  const int ks = tw_ksteps(cin_real);
  return ks > 0 && 16 * ks <= cin_ld && (cin_ld % 8) == 0;
}
size_t conv_tw_weight_elems(int cin_real) { return (size_t)2 * 3 * tw_ksteps(cin_real) * 64 * 8; }

hipError_t launch_pack_conv_tw(int dt, const void *w, int cin_real, int cin_ld, void *out, hipStream_t s) {
  const int ks = tw_ksteps(cin_real);
  if (dt == F32 || ks == 0) return hipErrorInvalidValue;
  if (dt == BF16) hipLaunchKernelGGL((pack_tw_kernel<bf16>), dim3(16), dim3(256), 0, s, static_cast<const bf16 *>(w), cin_ld, ks, static_cast<bf16 *>(out));
  else hipLaunchKernelGGL((pack_tw_kernel<f16>), dim3(16), dim3(256), 0, s, static_cast<const f16 *>(w), cin_ld, ks, static_cast<f16 *>(out));
  return hipGetLastError();
}

hipError_t launch_conv_tw(int dt, const void *in, int in_ld, int cin_real, const void *wfr, const float *bias, const void *res, int res_ld, void *out,
                          int out_ld, int N, int T, int HW, int relu, hipStream_t s) {
  const int ks = tw_ksteps(cin_real);
  if (dt == F32 || ks == 0 || N < 1 || T < 1 || HW < 1) return hipErrorInvalidValue;
  // The walk is HBM-bound and one workgroup occupies a CU: 25 tiles x 32 clips = 800 workgroups would run as four rounds with the last
  // one 12 % full.  Shrink the tile until the grid is a whole number of rounds (56 x 56 frames, 32 clips: 32 tiles of 98 positions).
  int tiles = (HW + TP - 1) / TP;
  {
    const long rounds = ((long)tiles * N + 255) / 256;
    const int want = (int)(rounds * 256 / N);
    if (want > tiles) tiles = want;
  }
  const int tp = std::min(TP, (HW + tiles - 1) / tiles);
  const dim3 grid((HW + tp - 1) / tp, N);
  const size_t lds = (size_t)3 * TP * (16 * ks + 8) * 2 + 64 * sizeof(float);
#define SF_TW(TT, KS_)                                                                                                                       \
  do {                                                                                                                                       \
    auto kern = conv_tw_kernel<TT, KS_>;                                                                                                     \
    static bool en = false;                                                                                                                  \
    if (!en) {                                                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);      \
      if (e != hipSuccess) return e;                                                                                                         \
      en = true;                                                                                                                             \
    }                                                                                                                                        \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, static_cast<const TT *>(in), in_ld, static_cast<const TT *>(wfr), bias,                \
                       static_cast<const TT *>(res), res_ld, static_cast<TT *>(out), out_ld, T, HW, relu, tp);                                   \
  } while (0)
  if (dt == BF16) {
    if (ks == 4) SF_TW(bf16, 4);
    else SF_TW(bf16, 10);
  } else {
    if (ks == 4) SF_TW(f16, 4);
    else SF_TW(f16, 10);
  }
#undef SF_TW
  return hipGetLastError();
}

}  // namespace sf
