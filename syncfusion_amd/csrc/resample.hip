// Band-limited sinc (Hann-windowed) polyphase resampler on the device -- SURVEY.md section 8f-2.
//
// Restates torchaudio.functional.resample (torchaudio==0.13.1, requirements.txt:19; defaults
// lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interpolation") that the reference calls on the
// CPU for every generated clip (main/generation.py:91-98, 48 kHz -> 22.05 kHz):
//   orig, new /= gcd;  base = min(orig,new)*rolloff;  width = ceil(lpw*orig/base)
//   bank[p][k] = sinc(pi t) * cos^2(pi t / (2 lpw)) * base/orig,  t = clamp((-p/new + (k-width)/orig) * base, +-lpw)
//   out[j*new + p] = sum_k bank[p][k] * xpad[j*orig + k]         (xpad = x padded with `width` / `width+orig` zeros)
//   truncated to ceil(new*L/orig) samples.
// The filter bank is built on the host in float64 and rounded to fp32 (as torchaudio does); the convolution is
// fp32.  One workgroup per output frame j: the (2*width+orig)-sample input window is staged in LDS once and every
// thread owns one phase p (its bank row streams from L2).
#include <cmath>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

__global__ __launch_bounds__(256) void resample_kernel(const float *__restrict__ x, int L, const float *__restrict__ bank, int orig,
                                                       int nnew, int width, int kw, float *__restrict__ out, int Lout) {
  extern __shared__ float win[];
  const int j = blockIdx.x, r = blockIdx.y;
  const float *xr = x + (size_t)r * L;
  const int base = j * orig - width;   // first input sample of the window (may be negative: left zero padding)
  for (int k = threadIdx.x; k < kw; k += blockDim.x) {
    const int i = base + k;
    win[k] = (i >= 0 && i < L) ? xr[i] : 0.f;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < nnew; p += blockDim.x) {
    const int o = j * nnew + p;
    if (o >= Lout) continue;
    const float *bp = bank + (size_t)p * kw;
    float acc = 0.f;
    for (int k = 0; k < kw; ++k) acc = fmaf(bp[k], win[k], acc);
    out[(size_t)r * Lout + o] = acc;
  }
}

}  // namespace

// host: filter bank [nnew][kw] as torchaudio builds it (float64 -> float32)
void resample_bank(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, std::vector<float> &bank, int &orig,
                   int &nnew, int &width) {
  int a = orig_freq, b = new_freq;
  while (b) {
    int t = a % b;
    a = b;
    b = t;
  }
  orig = orig_freq / a;
  nnew = new_freq / a;
  double base = (double)(orig < nnew ? orig : nnew) * rolloff;
  width = (int)std::ceil((double)lowpass_filter_width * orig / base);
  const int kw = 2 * width + orig;
  bank.assign((size_t)nnew * kw, 0.f);
  const double lpw = (double)lowpass_filter_width;
  const double scale = base / orig;
  for (int p = 0; p < nnew; ++p)
    for (int k = 0; k < kw; ++k) {
      // torchaudio 0.13.1 forms the phase offsets -p/new from an int64 arange divided by a Python int, i.e. in
      // float32, and only then adds the float64 sample positions: reproduce that rounding
      const double phase = (double)((float)(-p) / (float)nnew);
      double t = (phase + (double)(k - width) / orig) * base;
      if (t < -lpw) t = -lpw;
      if (t > lpw) t = lpw;
      const double c = std::cos(t * M_PI / lpw / 2.0);
      const double window = c * c;
      const double tp = t * M_PI;
      const double s = (tp == 0.0) ? 1.0 : std::sin(tp) / tp;
      bank[(size_t)p * kw + k] = (float)(s * (window * scale));
    }
}

hipError_t launch_resample(const float *x, int R, int L, const float *bank, int orig, int nnew, int width, float *out, int Lout,
                           hipStream_t s) {
  const int kw = 2 * width + orig;
  const int frames = (Lout + nnew - 1) / nnew;
  if ((size_t)kw * sizeof(float) > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(resample_kernel, dim3(frames, R), dim3(256), kw * sizeof(float), s, x, L, bank, orig, nnew, width, kw, out, Lout);
  return hipGetLastError();
}

}  // namespace sf
