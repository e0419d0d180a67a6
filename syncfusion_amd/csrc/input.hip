// Input side of the path, on the device:
//   * frames_preprocess: decoded uint8 RGB frames (N, T, H, W, 3) -> the onset net's input (N, 3, T, oh, ow) fp32, i.e. the
//     reference's ToTensor -> Resize((112, 112), antialias=True) -> Normalize(mean, std) -> (C, T, H, W) chain
//     (main/dataset_onset.py:47-50,152-165) in ONE pass over the pixels: the antialiased bilinear filter is ATen's separable
//     triangle filter (support = scale when downscaling, weights normalised per output pixel, align_corners = False);
//   * times_to_track: onset times (seconds) -> one-hot impulse track, `track[:, int(t * sr)] = 1` (main/dataset_diffusion.py:58-72),
//     evaluated in double precision like the Python expression.
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int kMaxTaps = 16;

struct AaAxis {
  int first, size;
  float w[kMaxTaps];
};
// ATen `_compute_indices_min_size_weights_aa` for one output index (bilinear: interp_size 2)
__device__ __forceinline__ AaAxis aa_axis(int i, int in_size, float scale) {
  AaAxis a;
  const float support = scale >= 1.0f ? scale : 1.0f;
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  const float center = scale * ((float)i + 0.5f);
  a.first = max((int)(center - support + 0.5f), 0);
  a.size = min(min((int)(center + support + 0.5f), in_size) - a.first, kMaxTaps);
  float tot = 0.f;
  for (int j = 0; j < kMaxTaps; ++j) {
    float w = 0.f;
    if (j < a.size) w = fmaxf(0.f, 1.0f - fabsf(((float)(j + a.first) - center + 0.5f) * invscale));
    a.w[j] = w;
    tot += w;
  }
  const float inv = tot != 0.f ? 1.0f / tot : 0.f;
  for (int j = 0; j < kMaxTaps; ++j) a.w[j] *= inv;
  return a;
}

__global__ void frames_preprocess_kernel(const unsigned char *__restrict__ fr, int NT, int T, int H, int W, int oh, int ow, float sy, float sx,
                                         float m0, float m1, float m2, float is0, float is1, float is2, float *__restrict__ out) {
  const int64_t total = (int64_t)NT * oh * ow;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % ow);
    const int64_t r = idx / ow;
    const int oy = (int)(r % oh);
    const int nt = (int)(r / oh);
    const AaAxis ay = aa_axis(oy, H, sy), ax = aa_axis(ox, W, sx);
    const unsigned char *img = fr + (size_t)nt * H * W * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int j = 0; j < ay.size; ++j) {
      const unsigned char *row = img + ((size_t)(ay.first + j) * W + ax.first) * 3;
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;   // horizontal pass first, like ATen's separable implementation
      for (int i = 0; i < ax.size; ++i) {
        const float w = ax.w[i];
        r0 = fmaf((float)row[3 * i + 0] * (1.0f / 255.0f), w, r0);
        r1 = fmaf((float)row[3 * i + 1] * (1.0f / 255.0f), w, r1);
        r2 = fmaf((float)row[3 * i + 2] * (1.0f / 255.0f), w, r2);
      }
      a0 = fmaf(r0, ay.w[j], a0);
      a1 = fmaf(r1, ay.w[j], a1);
      a2 = fmaf(r2, ay.w[j], a2);
    }
    const int n = nt / T, t = nt - n * T;
    const size_t plane = (size_t)oh * ow;
    float *o = out + (((size_t)n * 3) * T + t) * plane + (size_t)oy * ow + ox;
    o[0] = (a0 - m0) * is0;
    o[(size_t)T * plane] = (a1 - m1) * is1;
    o[2 * (size_t)T * plane] = (a2 - m2) * is2;
  }
}

__global__ void times_to_track_kernel(const double *__restrict__ times, const int *__restrict__ clip_of, int n_times, double sample_rate, int L,
                                      float *__restrict__ track) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_times) return;
  const long long pos = (long long)(times[i] * sample_rate);   // Python int(): truncation toward zero
  if (pos >= 0 && pos < L) track[(size_t)clip_of[i] * L + pos] = 1.0f;
}

}  // namespace

hipError_t launch_frames_preprocess(const unsigned char *frames, int N, int T, int H, int W, int oh, int ow, const float *mean, const float *stdv,
                                    float *out, hipStream_t s) {
  const float sy = (float)H / (float)oh, sx = (float)W / (float)ow;
  if (sy > (kMaxTaps - 1) / 2.0f || sx > (kMaxTaps - 1) / 2.0f) return hipErrorInvalidValue;   // filter wider than the tap table
  const int64_t total = (int64_t)N * T * oh * ow;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 65535);
  hipLaunchKernelGGL(frames_preprocess_kernel, dim3(grid), dim3(256), 0, s, frames, N * T, T, H, W, oh, ow, sy, sx, mean[0], mean[1], mean[2],
                     1.0f / stdv[0], 1.0f / stdv[1], 1.0f / stdv[2], out);
  return hipGetLastError();
}

hipError_t launch_times_to_track(const double *times, const int *clip_of, int n_times, double sample_rate, int B, int L, float *track, hipStream_t s) {
  hipError_t e = hipMemsetAsync(track, 0, (size_t)B * L * sizeof(float), s);
  if (e != hipSuccess || n_times == 0) return e;
  hipLaunchKernelGGL(times_to_track_kernel, dim3((n_times + 255) / 256), dim3(256), 0, s, times, clip_of, n_times, sample_rate, L, track);
  return hipGetLastError();
}

}  // namespace sf
