// Normalisation kernels (channels-last):
//   gn_stats     GroupNorm partial statistics  (mean, M2) per (clip, row-chunk, group); the consumer
//                convolution merges the chunks (Chan) in its prologue -- deterministic, no atomics.
//   ln_modulate  per-row LayerNorm over C fused with the a-unet Modulation  y = xhat*(1+s[b]) + t[b]
// Both are pure streaming kernels (HBM/L2-bound): 16-byte loads, wave shuffles, one pass.
#include "common.h"
#include "kernels.h"

namespace sf {

GnPlan gn_plan(int B, int L, int C) {
  GnPlan p;
  int nch0 = 512 / (B > 0 ? B : 1);
  nch0 = nch0 < 1 ? 1 : (nch0 > 32 ? 32 : nch0);
  int rows = (L + nch0 - 1) / nch0;
  int min_rows = 2048 / (C > 0 ? C : 1);
  if (min_rows < 1) min_rows = 1;
  if (rows < min_rows) rows = min_rows;
  if (rows > L) rows = L;
  p.chunk_rows = rows;
  p.nch = (L + rows - 1) / rows;
  return p;
}

namespace {

// One block = one (clip, chunk).  V = elements per access (16 bytes, or 1 for C < 16 bytes).
template <typename T, int V>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T *__restrict__ x, int ld, int L, int C, int G, int nch,
                                                       int chunk_rows, float *__restrict__ slab) {
  __shared__ float part_s[256 * 8];
  __shared__ float part_q[256 * 8];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / nch, ch = blockIdx.x - b * nch;
  const int r0 = ch * chunk_rows;
  const int rows = min(chunk_rows, L - r0);
  const int vpr = C / V;                     // accesses per row; host guarantees 256 % vpr == 0 or vpr % 256 == 0
  const T *base = x + ((size_t)b * L + r0) * ld;

  float s[V], q[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = q[j] = 0.f;

  if (vpr <= 256) {
    const int cv = tid % vpr;                // fixed column set of this thread
    const int rstep = 256 / vpr;
    for (int r = tid / vpr; r < rows; r += rstep) {
      const T *p = base + (size_t)r * ld + cv * V;
      if constexpr (V == 1) {
        float v = to_f(p[0]);
        s[0] += v;
        q[0] = fmaf(v, v, q[0]);
      } else {
        Vec16<T> v = ld16<T>(p);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float f = v.get(j);
          s[j] += f;
          q[j] = fmaf(f, f, q[j]);
        }
      }
    }
    // deterministic reduction: [row-slot][column] partials in LDS, summed in a fixed order per column
#pragma unroll
    for (int j = 0; j < V; ++j) {
      part_s[tid * V + j] = s[j];            // index = (tid/vpr)*C + cv*V + j  because tid = slot*vpr + cv
      part_q[tid * V + j] = q[j];
    }
    __syncthreads();
    const int slots = 256 / vpr;
    const int cpg = C / G;
    if (tid < G) {
      float ts = 0.f, tq = 0.f;
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c)
        for (int sl = 0; sl < slots; ++sl) {
          ts += part_s[sl * C + c];
          tq += part_q[sl * C + c];
        }
      float n = (float)rows * (float)cpg;
      float mean = ts / n;
      float m2 = fmaxf(tq - ts * mean, 0.f);
      float *o = slab + (((size_t)b * nch + ch) * G + tid) * 2;
      o[0] = mean;
      o[1] = m2;
    }
  } else {
    // wide rows (C/V > 256): every thread walks several column sets; group of a column set is uniform
    // per access because cpg >= V here.  Accumulate per group through LDS in a fixed order.
    const int cpg = C / G;
    for (int g = 0; g < G; ++g) {
      float ts = 0.f, tq = 0.f;
      const int v0 = g * cpg / V, v1 = (g + 1) * cpg / V;
      const int nv = v1 - v0;
      for (int i = tid; i < rows * nv; i += 256) {
        int r = i / nv, cv = v0 + (i - r * nv);
        Vec16<T> v = ld16<T>(base + (size_t)r * ld + cv * V);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float f = v.get(j);
          ts += f;
          tq = fmaf(f, f, tq);
        }
      }
      part_s[tid] = ts;
      part_q[tid] = tq;
      __syncthreads();
      if (tid == 0) {
        float a = 0.f, c = 0.f;
        for (int t = 0; t < 256; ++t) {
          a += part_s[t];
          c += part_q[t];
        }
        float n = (float)rows * (float)cpg;
        float mean = a / n;
        float *o = slab + (((size_t)b * nch + ch) * G + g) * 2;
        o[0] = mean;
        o[1] = fmaxf(c - a * mean, 0.f);
      }
      __syncthreads();
    }
  }
}

// TPR threads cooperate on one row; each holds VPT 16-byte vectors of it in registers.
template <typename T, int VPT>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const T *__restrict__ x, int ld, const float *__restrict__ ss,
                                                          int ss_ld, float eps, int rows, int L, int C, int tpr,
                                                          T *__restrict__ out, int out_ld) {
  constexpr int V = Vec16<T>::N;
  const int tid = threadIdx.x;
  const int rpb = 256 / tpr;
  const int row = blockIdx.x * rpb + tid / tpr;
  const int sub = tid % tpr;
  const bool active = row < rows;
  const int rr = active ? row : 0;
  Vec16<T> v[VPT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    v[i] = ld16<T>(x + (size_t)rr * ld + (size_t)(i * tpr + sub) * V);
#pragma unroll
    for (int j = 0; j < V; ++j) sum += v[i].get(j);
  }
  for (int o = tpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float d = v[i].get(j) - mean;
      sq = fmaf(d, d, sq);
    }
  for (int o = tpr >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = rsqrtf(sq / (float)C + eps);
  if (!active) return;
  const int b = row / L;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c0 = (i * tpr + sub) * V;
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float y = (v[i].get(j) - mean) * rstd;
      if (ss) y = fmaf(y, 1.0f + ss[(size_t)b * ss_ld + c0 + j], ss[(size_t)b * ss_ld + C + c0 + j]);
      o.set(j, y);
    }
    st16<T>(out + (size_t)row * out_ld + c0, o);
  }
}

template <typename T>
hipError_t ln_go(const void *x, int ld, const float *ss, int ss_ld, float eps, int B, int L, int C, void *out, int out_ld,
                 hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  if (C % V) return hipErrorInvalidValue;
  int vpr = C / V;
  int tpr = 1;
  while (tpr < 64 && tpr * 2 <= vpr) tpr *= 2;
  if (vpr % tpr) return hipErrorInvalidValue;
  int vpt = vpr / tpr;
  int rows = B * L;
  int rpb = 256 / tpr;
  dim3 grid((rows + rpb - 1) / rpb);
  const T *xp = static_cast<const T *>(x);
  T *op = static_cast<T *>(out);
  switch (vpt) {
    case 1: hipLaunchKernelGGL((ln_modulate_kernel<T, 1>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld); break;
    case 2: hipLaunchKernelGGL((ln_modulate_kernel<T, 2>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld); break;
    case 4: hipLaunchKernelGGL((ln_modulate_kernel<T, 4>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld); break;
    case 8: hipLaunchKernelGGL((ln_modulate_kernel<T, 8>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <typename T>
hipError_t gn_go(const void *x, int ld, int B, int L, int C, int G, int nch, int chunk_rows, float *slab, hipStream_t s) {
  constexpr int V = Vec16<T>::N;
  const T *xp = static_cast<const T *>(x);
  dim3 grid(B * nch);
  if (C % G) return hipErrorInvalidValue;
  if (C % V == 0) {
    int vpr = C / V;
    if (!((vpr <= 256 && 256 % vpr == 0) || (vpr > 256 && (C / G) % V == 0))) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gn_stats_kernel<T, V>), grid, dim3(256), 0, s, xp, ld, L, C, G, nch, chunk_rows, slab);
  } else {
    if (C > 256 || 256 % C) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gn_stats_kernel<T, 1>), grid, dim3(256), 0, s, xp, ld, L, C, G, nch, chunk_rows, slab);
  }
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gn_stats(int dt, const void *x, int ld, int B, int L, int C, int G, int nch, int chunk_rows, float *slab,
                           hipStream_t s) {
  return dt == F32 ? gn_go<float>(x, ld, B, L, C, G, nch, chunk_rows, slab, s)
                   : gn_go<bf16>(x, ld, B, L, C, G, nch, chunk_rows, slab, s);
}

hipError_t launch_ln_modulate(int dt, const void *x, int ld, const float *ss, int ss_ld, float eps, int B, int L, int C,
                              void *out, int out_ld, hipStream_t s) {
  return dt == F32 ? ln_go<float>(x, ld, ss, ss_ld, eps, B, L, C, out, out_ld, s)
                   : ln_go<bf16>(x, ld, ss, ss_ld, eps, B, L, C, out, out_ld, s);
}

}  // namespace sf
