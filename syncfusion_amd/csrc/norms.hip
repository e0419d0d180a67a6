// Normalisation kernels (channels-last):
//   gn_stats     GroupNorm partial statistics  (mean, M2) per (clip, row-chunk, group); the consumer
//                convolution merges the chunks (Chan) in its prologue -- deterministic, no atomics.
//   ln_modulate  per-row LayerNorm over C fused with the a-unet Modulation  y = xhat*(1+s[b]) + t[b]
// Both are pure streaming kernels (HBM/L2-bound): 16-byte loads, wave shuffles, one pass.
#include <cstdlib>
#include <type_traits>

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

// largest number of 16-byte vectors per thread the register-resident GroupNorm+SiLU kernel is used for (tuning hook SF_GN_REG_MAXV:
// 4 / 8 / 16); shared by gn_silu_go and gn_silu_ws_go so that an A/B with the hook compares exactly two paths
int gn_reg_max_rv() {
  static const int v = [] {
    const char *e = tune_env("SF_GN_REG_MAXV");
    const int x = e ? atoi(e) : 16;
    return x >= 16 ? 16 : (x >= 8 ? 8 : 4);
  }();
  return v;
}

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
    if ((G & (G - 1)) == 0 && G <= 256) {
      // 256/G threads per group: strided partial sums, then a fixed-shape LDS tree (deterministic)
      const int tpg = 256 / G, g = tid / tpg, j = tid - g * tpg;
      const int ne = cpg * slots;
      float ts = 0.f, tq = 0.f;
      for (int e = j; e < ne; e += tpg) {
        const int sl = e / cpg, c = g * cpg + (e - sl * cpg);
        ts += part_s[sl * C + c];
        tq += part_q[sl * C + c];
      }
      __syncthreads();
      part_s[tid] = ts;
      part_q[tid] = tq;
      __syncthreads();
      for (int off = tpg >> 1; off > 0; off >>= 1) {
        if (j < off) {
          part_s[tid] += part_s[tid + off];
          part_q[tid] += part_q[tid + off];
        }
        __syncthreads();
      }
      if (j == 0) {
        const float a = part_s[tid], q2 = part_q[tid];
        const float n = (float)rows * (float)cpg;
        const float mean = a / n;
        float *o = slab + (((size_t)b * nch + ch) * G + g) * 2;
        o[0] = mean;
        o[1] = fmaxf(q2 - a * mean, 0.f);
      }
    } else if (tid < G) {
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
                                                          T *__restrict__ out, int out_ld, const int nreal, const Prefetch pf, const int xfmt) {
  constexpr int V = Vec16<T>::N;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= nreal) {   // hosted weight prefetch for the GEMM that follows (kernels.h, Prefetch)
    prefetch_slice(pf, (int)blockIdx.x - nreal, 256);
    return;
  }
  const int rpb = 256 / tpr;
  const int row = blockIdx.x * rpb + tid / tpr;
  const int sub = tid % tpr;
  const bool active = row < rows;
  const int rr = active ? row : 0;
  Vec16<T> v[VPT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) v[i] = ld16<T>(x + (size_t)rr * ld + (size_t)(i * tpr + sub) * V);
  // the per-clip scale / shift do not depend on the statistics: issue their loads now, next to the row's, so the
  // kernel pays ONE memory round trip instead of two (16-byte loads; the engine keeps every offset 16-byte aligned)
  const int b = rr / L;
  f32x4 sc[VPT][V / 4], sh[VPT][V / 4];
  if (ss) {
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
      for (int q = 0; q < V / 4; ++q) {
        const int c0 = (i * tpr + sub) * V + 4 * q;
        sc[i][q] = *reinterpret_cast<const f32x4 *>(ss + (size_t)b * ss_ld + c0);
        sh[i][q] = *reinterpret_cast<const f32x4 *>(ss + (size_t)b * ss_ld + C + c0);
      }
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i)
#pragma unroll
    for (int j = 0; j < V; ++j) sum += v[i].get(j);
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
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c0 = (i * tpr + sub) * V;
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float y = (v[i].get(j) - mean) * rstd;
      if (ss) y = fmaf(y, 1.0f + sc[i][j >> 2][j & 3], sh[i][j >> 2][j & 3]);
      o.set(j, y);
    }
    if constexpr (sizeof(T) == 4) {
      if (xfmt) {   // pre-split rows for the GEMM that follows (common.h, st4_x3)
        st4_x3(out + (size_t)row * out_ld, c0, o.v);
        continue;
      }
    }
    st16<T>(out + (size_t)row * out_ld + c0, o);
  }
}

// GroupNorm + SiLU materialised once:  y = silu((x - mean_bg) * rstd_bg * gamma_c + beta_c).
// One workgroup per (clip, group): pass 1 reduces the group's L x (C/G) slab (pivot-shifted sum / sum of
// squares, fixed-shape LDS tree -> deterministic), pass 2 re-reads it (L2-resident: a slab is 11-45 K
// elements at every level of the reference U-Net) and writes the activated tensor.  Used in front of every
// wide convolution: applying the activation in the GEMM's A-load instead would repeat the exp/rcp for every
// column tile and tap (measured: 40 us instead of 5 us per layer at C = 1024).
template <typename T, int VW>
__global__ __launch_bounds__(512) void gn_silu_kernel(const T *__restrict__ x, int ld, int L, int C, int G,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                      T *__restrict__ out, int out_ld, const int xfmt) {
  constexpr bool FAST = sizeof(T) == 2;
  __shared__ float red_s[512], red_q[512];
  __shared__ float stat[2];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int cpg = C / G;
  const int vr = cpg / VW;                 // accesses per row
  const int nv = L * vr;
  const T *base = x + (size_t)b * L * ld + (size_t)g * cpg;
  T *obase = out + (size_t)b * L * out_ld + (size_t)g * cpg;
  const float pivot = to_f(base[0]);
  float s = 0.f, q = 0.f;
#pragma unroll 4
  for (int i = tid; i < nv; i += 512) {
    const int r = i / vr, cv = i - r * vr;
    T v[VW];
    __builtin_memcpy(v, __builtin_assume_aligned(base + (size_t)r * ld + cv * VW, VW * sizeof(T)), VW * sizeof(T));
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float d = to_f(v[j]) - pivot;
      s += d;
      q = fmaf(d, d, q);
    }
  }
  red_s[tid] = s;
  red_q[tid] = q;
  __syncthreads();
  for (int off = 256; off > 0; off >>= 1) {
    if (tid < off) {
      red_s[tid] += red_s[tid + off];
      red_q[tid] += red_q[tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float n = (float)L * (float)cpg;
    const float md = red_s[0] / n;
    const float var = fmaxf(red_q[0] / n - md * md, 0.f);
    stat[0] = pivot + md;
    stat[1] = rsqrtf(var + eps);
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
#pragma unroll 4
  for (int i = tid; i < nv; i += 512) {
    const int r = i / vr, cv = i - r * vr;
    T v[VW], o[VW];
    __builtin_memcpy(v, __builtin_assume_aligned(base + (size_t)r * ld + cv * VW, VW * sizeof(T)), VW * sizeof(T));
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int c = g * cpg + cv * VW + j;
      const float sc = rstd * gamma[c];
      o[j] = from_f<T>(silu_t<FAST>(fmaf(to_f(v[j]) - mean, sc, beta[c])));
    }
    if constexpr (sizeof(T) == 4 && VW == 4) {
      if (xfmt) {   // pre-split rows for the GEMM that follows (common.h, st4_x3)
        st4_x3(out + ((size_t)b * L + r) * out_ld, g * cpg + cv * VW, f32x4{o[0], o[1], o[2], o[3]});
        continue;
      }
    }
    __builtin_memcpy(__builtin_assume_aligned(obase + (size_t)r * out_ld + cv * VW, VW * sizeof(T)), o, VW * sizeof(T));
  }
}

// Same operation when the (clip, group) slab fits the workgroup's registers (<= RV 16-byte vectors per thread: every level
// of the reference U-Net at 2 s clips): x, gamma and beta are fetched in ONE memory round trip, the statistics are
// two in-register passes (sum -> mean, centred squares -> variance) with a DPP wave reduction and one barrier each.
template <typename T, int RV>
__global__ __launch_bounds__(512) void gn_silu_reg_kernel(const T *__restrict__ x, int ld, int L, int C, int G,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                          T *__restrict__ out, int out_ld, const int nslab, const Prefetch pf, const int xfmt) {
  constexpr bool FAST = sizeof(T) == 2;
  constexpr int V = Vec16<T>::N;
  __shared__ float red[2][8];
  if ((int)blockIdx.x >= nslab) {   // hosted weight prefetch for the convolution that follows (kernels.h, Prefetch)
    prefetch_slice(pf, (int)blockIdx.x - nslab, 512);
    return;
  }
  const int tid = threadIdx.x, wave = tid >> 6;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int cpg = C / G;
  const int vr = cpg / V;                  // vectors per row
  const int nv = L * vr;
  const T *base = x + (size_t)b * L * ld + (size_t)g * cpg;
  T *obase = out + (size_t)b * L * out_ld + (size_t)g * cpg;
  // the launcher guarantees 512 % vr == 0: every vector of a thread sits in the same channel octet (one gamma / beta fetch)
  Vec16<T> v[RV];
  const int cv = tid % vr;
  f32x4 ga[V / 4], be[V / 4];
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    ga[q] = *reinterpret_cast<const f32x4 *>(gamma + g * cpg + cv * V + 4 * q);
    be[q] = *reinterpret_cast<const f32x4 *>(beta + g * cpg + cv * V + 4 * q);
  }
  int rr[RV];
#pragma unroll
  for (int i = 0; i < RV; ++i) {
    const int idx = tid + i * 512;
    const bool on = idx < nv;
    rr[i] = on ? idx / vr : 0;
    v[i] = on ? ld16<T>(base + (size_t)rr[i] * ld + cv * V) : zero16<T>();
  }
  const float n = (float)L * (float)cpg;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RV; ++i)
#pragma unroll
    for (int j = 0; j < V; ++j) s += v[i].get(j);   // padding vectors are zero
  s = wave_sum_dpp(s);
  if ((tid & 63) == 0) red[0][wave] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += red[0][w];
  const float mean = tot / n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < RV; ++i) {
    if (tid + i * 512 < nv) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d = v[i].get(j) - mean;
        q = fmaf(d, d, q);
      }
    }
  }
  q = wave_sum_dpp(q);
  if ((tid & 63) == 0) red[1][wave] = q;
  __syncthreads();
  float tq = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tq += red[1][w];
  const float rstd = rsqrtf(tq / n + eps);
  float sc[V], sh[V];   // y = x * sc + sh
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sc[j] = rstd * ga[j >> 2][j & 3];
    sh[j] = fmaf(-mean, sc[j], be[j >> 2][j & 3]);
  }
#pragma unroll
  for (int i = 0; i < RV; ++i) {
    if (tid + i * 512 < nv) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.set(j, silu_t<FAST>(fmaf(v[i].get(j), sc[j], sh[j])));
      if constexpr (sizeof(T) == 4) {
        if (xfmt) {
          st4_x3(out + ((size_t)b * L + rr[i]) * out_ld, g * cpg + cv * V, o.v);
          continue;
        }
      }
      st16<T>(obase + (size_t)rr[i] * out_ld + cv * V, o);
    }
  }
}

// Long sequences (the reference's 2^18-sample clips: 65 K elements per (clip, group) at depths 4-6): the one-workgroup-per-slab
// kernel above leaves most CUs idle (80 workgroups at ten clips per branch) and walks each slab twice.  With the chunk statistics of
// gn_stats in hand, the activation is ONE streaming pass over (clip, row chunk) workgroups that cover all channels: a thread keeps
// its V channels (C / V divides 256), merges the chunk partials of their group (Chan, chunk order) once, then streams rows.
template <typename T>
__global__ __launch_bounds__(256) void gn_silu_apply_kernel(const T *__restrict__ x, int ld, int L, int C, int G, const float *__restrict__ slab,
                                                            int nch, int chunk_rows, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, int rows_per_wg, T *__restrict__ out,
                                                            int out_ld, const int xfmt = 0) {
  constexpr bool FAST = sizeof(T) == 2;
  constexpr int V = Vec16<T>::N;
  __shared__ float mean_s[64], rstd_s[64];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int cpg = C / G;
  for (int g = tid; g < G; g += 256) {
    const float *sb = slab + ((size_t)b * nch * G + g) * 2;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int ch0 = 0; ch0 < nch; ch0 += 8) {   // eight chunk partials per memory round trip, merged in chunk order (same bits as one at a time)
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ch0 + u < nch ? *reinterpret_cast<const float2 *>(sb + (size_t)(ch0 + u) * G * 2) : make_float2(0.f, 0.f);
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
  __syncthreads();
  const int vpr = C / V, cv = tid % vpr, rstep = 256 / vpr;
  float sc[V], sh[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = cv * V + j, g = c / cpg;
    sc[j] = rstd_s[g] * gamma[c];
    sh[j] = fmaf(-mean_s[g], sc[j], beta[c]);
  }
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(L, r0 + rows_per_wg);
  const T *xb = x + ((size_t)b * L) * ld + cv * V;
  T *ob = out + ((size_t)b * L) * out_ld + cv * V;
#pragma unroll 4
  for (int r = r0 + tid / vpr; r < r1; r += rstep) {
    const Vec16<T> v = ld16<T>(xb + (size_t)r * ld);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.set(j, silu_t<FAST>(fmaf(v.get(j), sc[j], sh[j])));
    if constexpr (sizeof(T) == 4) {
      if (xfmt) {
        st4_x3(out + ((size_t)b * L + r) * out_ld, cv * V, o.v);
        continue;
      }
    }
    st16<T>(ob + (size_t)r * out_ld, o);
  }
}

template <typename T>
hipError_t gn_silu_go(const void *x, int ld, int B, int L, int C, int G, const float *gamma, const float *beta, float eps, void *out,
                      int out_ld, hipStream_t s, Prefetch pf = Prefetch(), bool xfmt = false) {
  if (C % G) return hipErrorInvalidValue;
  const int xf = xfmt ? 1 : 0;
  if (xfmt && (sizeof(T) != 4 || (C % 32) || out_ld != C || (ld % 4) || ((C / G) % 4))) return hipErrorInvalidValue;
  const int cpg = C / G;
  constexpr int V = Vec16<T>::N;
  const T *xp = static_cast<const T *>(x);
  T *op = static_cast<T *>(out);
  dim3 grid(B * G);
#define SF_GNS(VW) hipLaunchKernelGGL((gn_silu_kernel<T, VW>), grid, dim3(512), 0, s, xp, ld, L, C, G, gamma, beta, eps, op, out_ld, xf)
  const bool al = (ld % V == 0) && (out_ld % V == 0);
  // slab fits the registers of one workgroup: up to 16 vectors of 16 bytes per thread (64 data registers of the 256 a wave of a
  // 512-thread workgroup may hold).  8 and 16 cover the 2^18-sample clips (65 K elements per slab at depths 3-6), which the two-pass
  // kernel below walked twice with four loads in flight: 16.4 us per launch, 10.6 % of that step (profiles/r3_h_refshape_*).
  const int max_rv = gn_reg_max_rv();
  if (al && cpg % V == 0 && 512 % (cpg / V) == 0 && (int64_t)L * (cpg / V) <= (int64_t)max_rv * 512) {
    const int64_t nv = (int64_t)L * (cpg / V);
    const int nslab = B * G;
    const dim3 gridp(nslab + (pf.ptr && pf.bytes >= 16 ? pf.wgs : 0));
#define SF_GNR(RV) hipLaunchKernelGGL((gn_silu_reg_kernel<T, RV>), gridp, dim3(512), 0, s, xp, ld, L, C, G, gamma, beta, eps, op, out_ld, nslab, pf, xf)
    if (nv <= 2 * 512) SF_GNR(2);
    else if (nv <= 4 * 512) SF_GNR(4);
    else if (nv <= 8 * 512) SF_GNR(8);
    else SF_GNR(16);
#undef SF_GNR
    return hipGetLastError();
  }
  if (al && cpg % V == 0) SF_GNS(V);
  else if (ld % 4 == 0 && out_ld % 4 == 0 && cpg % 4 == 0) SF_GNS(4);
  else if (ld % 2 == 0 && out_ld % 2 == 0 && cpg % 2 == 0) SF_GNS(2);
  else SF_GNS(1);
#undef SF_GNS
  return hipGetLastError();
}

template <typename T>
hipError_t ln_go(const void *x, int ld, const float *ss, int ss_ld, float eps, int B, int L, int C, void *out, int out_ld,
                 hipStream_t s, Prefetch pf = Prefetch(), bool xfmt = false) {
  constexpr int V = Vec16<T>::N;
  if (C % V) return hipErrorInvalidValue;
  if (xfmt && (sizeof(T) != 4 || (C % 32) || out_ld != C)) return hipErrorInvalidValue;
  const int xf = xfmt ? 1 : 0;
  if (ss && ((ss_ld % 4) || (reinterpret_cast<uintptr_t>(ss) % 16))) return hipErrorInvalidValue;   // 16-byte scale/shift loads
  int vpr = C / V;
  int tpr = 1;
  while (tpr < 64 && tpr * 2 <= vpr) tpr *= 2;
  if (vpr % tpr) return hipErrorInvalidValue;
  int vpt = vpr / tpr;
  int rows = B * L;
  int rpb = 256 / tpr;
  const int nreal = (rows + rpb - 1) / rpb;
  dim3 grid(nreal + (pf.ptr && pf.bytes >= 16 ? pf.wgs : 0));
  const T *xp = static_cast<const T *>(x);
  T *op = static_cast<T *>(out);
  switch (vpt) {
    case 1: hipLaunchKernelGGL((ln_modulate_kernel<T, 1>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld, nreal, pf, xf); break;
    case 2: hipLaunchKernelGGL((ln_modulate_kernel<T, 2>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld, nreal, pf, xf); break;
    case 4: hipLaunchKernelGGL((ln_modulate_kernel<T, 4>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld, nreal, pf, xf); break;
    case 8: hipLaunchKernelGGL((ln_modulate_kernel<T, 8>), grid, dim3(256), 0, s, xp, ld, ss, ss_ld, eps, rows, L, C, tpr, op, out_ld, nreal, pf, xf); break;
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
  return SF_DISPATCH_T(dt, gn_go<T>(x, ld, B, L, C, G, nch, chunk_rows, slab, s));
}

// The same with a statistics slab at hand (>= B * 32 * G * 2 floats): slabs that do not fit one workgroup's registers take the
// chunked two-launch path (gn_stats + gn_silu_apply); everything else goes to launch_gn_silu.
template <typename T>
static hipError_t gn_silu_ws_go(const void *x, int ld, int B, int L, int C, int G, const float *gamma, const float *beta, float eps, void *out,
                                int out_ld, float *slab, int64_t slab_floats, hipStream_t s, bool &done) {
  constexpr int V = Vec16<T>::N;
  done = false;
  if (C % G || G > 64 || C % V || (256 % (C / V)) || (ld % V) || (out_ld % V)) return hipSuccess;
  const int cpg = C / G;
  if (cpg % V == 0 && 512 % (cpg / V) == 0 && (int64_t)L * (cpg / V) <= (int64_t)gn_reg_max_rv() * 512) return hipSuccess;   // register-resident kernel applies
  if ((int64_t)L * cpg < 32768) return hipSuccess;                                                       // short slabs: one launch wins
  const GnPlan gp = gn_plan(B, L, C);
  if ((int64_t)B * gp.nch * G * 2 > slab_floats) return hipSuccess;
  hipError_t e = launch_gn_stats(std::is_same<T, float>::value ? F32 : (std::is_same<T, f16>::value ? F16 : BF16), x, ld, B, L, C, G, gp.nch,
                                 gp.chunk_rows, slab, s);
  if (e != hipSuccess) return e;
  int nwg = (int)std::min<int64_t>(64, std::max<int64_t>(1, ((int64_t)L * C + 32767) / 32768));   // >= 32 K elements per workgroup
  const int rows_per_wg = (L + nwg - 1) / nwg;
  nwg = (L + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL((gn_silu_apply_kernel<T>), dim3(nwg, B), dim3(256), 0, s, static_cast<const T *>(x), ld, L, C, G, slab, gp.nch, gp.chunk_rows, gamma,
                     beta, eps, rows_per_wg, static_cast<T *>(out), out_ld);
  done = true;
  return hipGetLastError();
}

hipError_t launch_gn_silu_ws(int dt, const void *x, int ld, int B, int L, int C, int G, const float *gamma, const float *beta, float eps, void *out,
                             int out_ld, float *slab, int64_t slab_floats, hipStream_t s) {
  static const bool off = tune_env("SF_NO_GN_CHUNKED") != nullptr;   // tuning hook
  if (slab && !off) {
    bool done = false;
    hipError_t e = SF_DISPATCH_T(dt, gn_silu_ws_go<T>(x, ld, B, L, C, G, gamma, beta, eps, out, out_ld, slab, slab_floats, s, done));
    if (e != hipSuccess || done) return e;
  }
  return launch_gn_silu(dt, x, ld, B, L, C, G, gamma, beta, eps, out, out_ld, s);
}

hipError_t launch_gn_silu(int dt, const void *x, int ld, int B, int L, int C, int G, const float *gamma, const float *beta, float eps,
                          void *out, int out_ld, hipStream_t s, Prefetch pf, bool xfmt) {
  if (xfmt && dt != F32) return hipErrorInvalidValue;
  return SF_DISPATCH_T(dt, gn_silu_go<T>(x, ld, B, L, C, G, gamma, beta, eps, out, out_ld, s, pf, xfmt));
}

hipError_t launch_ln_modulate(int dt, const void *x, int ld, const float *ss, int ss_ld, float eps, int B, int L, int C,
                              void *out, int out_ld, hipStream_t s, Prefetch pf, bool xfmt) {
  if (xfmt && dt != F32) return hipErrorInvalidValue;
  return SF_DISPATCH_T(dt, ln_go<T>(x, ld, ss, ss_ld, eps, B, L, C, out, out_ld, s, pf, xfmt));
}

}  // namespace sf
