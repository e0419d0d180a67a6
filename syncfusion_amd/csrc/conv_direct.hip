// Direct (VALU) 1-D convolution for thin layers: the U-Net's depth-0 (C = 8), its 1->8 entry and
// 8->1 exit convolutions, and the shallow levels of Encoder1d (C = 1..32).  These layers have a few
// hundred FLOPs per output row and are bound by streaming the activations, so there is nothing for
// the matrix cores to do: one thread owns one output row (all N <= 32 output channels in registers),
// lanes walk consecutive rows (coalesced channels-last reads), and the weights sit in LDS as
// [k][N] so every lane reads the same address (broadcast, conflict-free).
//
// Same semantics as conv_gemm (ConvGemmArgs): optional GroupNorm+SiLU prologue from the partial
// statistics slab, channel concat of a second source, bias / per-clip scale / residual / per-clip
// add / relu epilogue.
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

template <typename TI, typename TO, int NT>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvGemmArgs a) {
  constexpr bool FAST = sizeof(TI) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *wl = reinterpret_cast<float *>(smem);                      // [K][NT]
  float2 *tab = reinterpret_cast<float2 *>(wl + (size_t)a.K * NT);  // GN (scale, shift) per (clip, channel)

  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 256;
  const float *w = static_cast<const float *>(a.w);
  for (int idx = tid; idx < a.K * NT; idx += 256) {
    int k = idx / NT, n = idx - k * NT;
    wl[idx] = n < a.N ? w[(size_t)n * a.K + k] : 0.f;
  }
  const int b_first = m0 / a.Lout;
  if (a.pro == 1) {
    const int b_last = min(a.M - 1, m0 + 255) / a.Lout;
    const int nb = b_last - b_first + 1;
    float2 *mr = tab + (size_t)a.cin * nb;
    const int cpg = a.cin / a.G;
    for (int idx = tid >> 5; idx < nb * a.G; idx += 8) {   // one half-wave per (clip, group)
      const int bl = idx / a.G, g = idx - bl * a.G;
      const float *sl = a.stats + ((size_t)(b_first + bl) * a.nch) * a.G * 2 + g * 2;
      const float2 r = gn_merge32(sl, a.G, a.nch, a.chunk_rows, a.Lsrc, cpg, a.eps, tid & 31);
      if ((tid & 31) == 0) mr[idx] = r;
    }
    __syncthreads();
    for (int idx = tid; idx < nb * a.cin; idx += 256) {
      int bl = idx / a.cin, c = idx - bl * a.cin;
      float2 s = mr[bl * a.G + c / cpg];
      float sc = s.y * a.gamma[c];
      tab[idx] = make_float2(sc, a.beta[c] - s.x * sc);
    }
  }
  __syncthreads();

  const int m = m0 + tid;
  if (m >= a.M) return;
  const int b = m / a.Lout;
  const int l = m - b * a.Lout;
  const TI *src = static_cast<const TI *>(a.src);
  const TI *src2 = static_cast<const TI *>(a.src2);

  float acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = 0.f;

  const float2 *tb = tab + (size_t)(b - b_first) * a.cin;
  constexpr int VI = Vec16<TI>::N;
  const bool vec_ok = (a.cin % VI) == 0 && (a.src_ld % VI) == 0;
  const int pmax = (a.Lsrc << a.up_shift) - 1;
  for (int tap = 0; tap < a.taps; ++tap) {
    // unconditional loads from a clamped row; rows in the zero padding contribute through `live` = 0
    const int p = l * a.stride + tap - a.pad;
    const float live = (p >= 0 && p <= pmax) ? 1.f : 0.f;
    const TI *row = src + (size_t)(b * a.Lsrc + (min(max(p, 0), pmax) >> a.up_shift)) * a.src_ld;
    const float *wk = wl + (size_t)tap * a.cin * NT;
    if (vec_ok) {
      for (int c0 = 0; c0 < a.cin; c0 += VI) {
        const Vec16<TI> v = ld16<TI>(row + c0);
#pragma unroll
        for (int j = 0; j < VI; ++j) {
          float x = v.get(j);
          if (a.pro == 1) {
            const float2 sd = tb[c0 + j];
            x = silu_t<FAST>(fmaf(x, sd.x, sd.y));
          }
          x *= live;
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = fmaf(x, wk[(c0 + j) * NT + n], acc[n]);
        }
      }
    } else {
      for (int ci = 0; ci < a.cin; ++ci) {
        float x = to_f(row[ci]);
        if (a.pro == 1) {
          const float2 sd = tb[ci];
          x = silu_t<FAST>(fmaf(x, sd.x, sd.y));
        }
        x *= live;
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = fmaf(x, wk[ci * NT + n], acc[n]);
      }
    }
  }
  if (a.cin2 > 0) {
    const TI *row = src2 + (size_t)m * a.src2_ld;
    const float *wk = wl + (size_t)a.taps * a.cin * NT;
    for (int ci = 0; ci < a.cin2; ++ci) {
      float x = to_f(row[ci]);
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = fmaf(x, wk[ci * NT + n], acc[n]);
    }
  }

  // epilogue: operand loads batched and unconditional, one predicated store pass
  TO *out = static_cast<TO *>(a.out) + (size_t)m * a.out_ld;
  const TO *res = static_cast<const TO *>(a.res);
  const bool has_res = res != nullptr, has_bs = a.bscale != nullptr, has_ba = a.badd != nullptr;
  float bi[NT], rv[NT], sv[NT], av[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int nc = min(n, a.N - 1);
    bi[n] = a.bias ? a.bias[nc] : 0.f;
    rv[n] = has_res ? to_f(res[(size_t)m * a.res_ld + nc]) : 0.f;
    sv[n] = has_bs ? a.bscale[(size_t)b * a.bscale_ld + nc] : 1.f;
    av[n] = has_ba ? a.badd[(size_t)b * a.badd_ld + nc] : 0.f;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float v = (acc[n] + bi[n]) * sv[n] + rv[n] + av[n];
    v = n < a.N ? apply_act(v, a.act) : 0.f;
    if (n < a.n_store) out[n] = from_f<TO>(v);
  }
}

template <typename TI, typename TO> hipError_t go(const ConvGemmArgs &a, hipStream_t s) {
  size_t lds = 0;
  auto lds_for = [&](int nt) {
    size_t b = (size_t)a.K * nt * sizeof(float);
    if (a.pro == 1) {
      int nb = min(a.M / a.Lout + 1, 256 / a.Lout + 2);
      b += (size_t)nb * (a.cin + a.G) * sizeof(float2);
    }
    return b;
  };
  dim3 grid((a.M + 255) / 256);
#define SF_GO(NT)                                                                                    \
  {                                                                                                  \
    lds = lds_for(NT);                                                                               \
    if (lds > 64 * 1024) return hipErrorInvalidValue;                                                \
    auto kern = conv_direct_kernel<TI, TO, NT>;                                                      \
    static bool en = false;                                                                          \
    if (!en) {                                                                                       \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);     \
      if (e != hipSuccess) return e;                                                                 \
      en = true;                                                                                     \
    }                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);                                            \
    return hipGetLastError();                                                                        \
  }
  const int ns = a.n_store;
  if (ns <= 1) SF_GO(1)
  if (ns <= 2) SF_GO(2)
  if (ns <= 4) SF_GO(4)
  if (ns <= 8) SF_GO(8)
  if (ns <= 16) SF_GO(16)
  if (ns <= 32) SF_GO(32)
#undef SF_GO
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_conv_direct(int dt_in, int dt_out, const ConvGemmArgs &a, hipStream_t s) {
  if (a.geom != 0 || a.M <= 0) return hipErrorInvalidValue;
  if (a.pro == 1 && (a.cin % a.G)) return hipErrorInvalidValue;
  if (dt_in == F32 && dt_out == F32) return go<float, float>(a, s);
  if (dt_in == F32 && dt_out == BF16) return go<float, bf16>(a, s);
  if (dt_in == BF16 && dt_out == BF16) return go<bf16, bf16>(a, s);
  if (dt_in == BF16 && dt_out == F32) return go<bf16, float>(a, s);
  if (dt_in == F32 && dt_out == F16) return go<float, f16>(a, s);
  if (dt_in == F16 && dt_out == F16) return go<f16, f16>(a, s);
  if (dt_in == F16 && dt_out == F32) return go<f16, float>(a, s);
  return hipErrorInvalidValue;
}

}  // namespace sf
