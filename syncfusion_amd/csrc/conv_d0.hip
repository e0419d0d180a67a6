// The 8-channel level of the U-Net (depth 0 of the reference config: C = 8, one channel per GroupNorm group, L = L0 positions per
// clip) on the VECTOR units, one position per lane.
//
// Why not conv_thin's MFMA formulation here: a 32x32x16 tile at C = 8 uses 8 of its 32 output columns and 24 of 32 reduction
// slots, and the per-tile bookkeeping (fragment shuffles, epilogue selects, statistics) is ~1000 vector instructions for 32
// positions x 8 channels = 256 outputs.  A lane that owns ONE position needs 8 x 24 = 192 FMAs for the convolution and a few dozen
// instructions for everything else: about a quarter of the instructions per position, and instruction issue -- not HBM -- is what
// bounds this level (the MFMA kernels run at 5-6x the time their 16 bytes per position and tensor would take to stream).
//
//   d0_conv:  out = Conv3(SiLU(GroupNorm(x))) + bias (* bscale) (+ res) (+ badd);  GroupNorm partial of the stored output
//   d0_tail:  y = x + Conv3(SiLU(GroupNorm(h))) + b2;  m = LN_8(y) (1 + s) + t;  out = m + W3 [m | ctx] + b3 (+ badd);  partial
// Same argument blocks, chunking and statistics layout as conv_thin / thin_tail (kernels.h): a drop-in for C = N = 8.
//
// Layout of a pass: a wave covers 62 consecutive positions; lane l holds position base + l - 1, so lanes 0 and 63 are the halo
// whose activation only feeds their neighbours (DPP wave shifts) -- every activation is computed once (+ 3 %), never per tap.
//
// Measured (tools/d0_bench.hip, 64 clips x 11264 positions, bf16): conv 16.3 us, tail 20.5 us at 992 positions per workgroup
// (conv_thin / thin_tail on the same tensors: ~30 / ~40 us).  The kernels are VALU-throughput bound (~330 vector instructions per
// pass, 16 of them quarter-rate exp / rcp): without the convolution FMAs the conv kernel still takes 12 us, and more waves per
// workgroup or per SIMD do not help (profiles/r3_j_d0_sweep.txt).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int CH = 8;
constexpr int SPAN = 62;   // positions a wave owns per pass

// eight channels of one position as loaded (16 bytes in the 16-bit types): what a pass keeps in flight for the NEXT pass
template <typename T> struct Raw8 {
  Vec16<T> v[sizeof(T) == 2 ? 1 : 2];
  __device__ __forceinline__ void load(const T *p) {
    v[0] = ld16<T>(p);
    if constexpr (sizeof(T) == 4) v[1] = ld16<T>(p + 4);
  }
  __device__ __forceinline__ void zero() {
    v[0] = zero16<T>();
    if constexpr (sizeof(T) == 4) v[1] = zero16<T>();
  }
  __device__ __forceinline__ void get(float (&o)[CH]) const {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] = v[0].get(j);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = v[0].get(j);
        o[4 + j] = v[1].get(j);
      }
    }
  }
};
// rounds v to T in place (the statistics see what is stored)
template <typename T> __device__ __forceinline__ void round8(float (&v)[CH]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int j = 0; j < CH; ++j) v[j] = to_f(from_f<T>(v[j]));
  }
}
template <typename T> __device__ __forceinline__ void store8(T *p, const float (&v)[CH]) {
  if constexpr (sizeof(T) == 2) {
    Vec16<T> r;
#pragma unroll
    for (int j = 0; j < CH; ++j) r.set(j, v[j]);
    st16<T>(p, r);
  } else {
    *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4 *>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}

// (mean, rstd) of GroupNorm group g (= channel g: one channel per group) of clip b from its chunk partials; one half-wave per call,
// every lane of the half-wave returns the result.  Same merge order as conv_thin's gn_merge_n (deterministic).
template <bool FAST>
__device__ __forceinline__ float2 merge_partials(const float *__restrict__ sl, int G, int nch, int chunk_rows, int L, float eps, int lane32) {
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int i = lane32; i < nch; i += 32) {
    const int rows = min(chunk_rows, L - i * chunk_rows);
    welford_merge_t<FAST>(n, mean, m2, (float)rows, sl[(size_t)i * G * 2], sl[(size_t)i * G * 2 + 1]);
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const float nb = __shfl_down(n, off, 32), mb = __shfl_down(mean, off, 32), qb = __shfl_down(m2, off, 32);
    welford_merge_t<FAST>(n, mean, m2, nb, mb, qb);
  }
  const float mu = __shfl(mean, 0, 32), var = div_t<FAST>(__shfl(m2, 0, 32), __shfl(n, 0, 32));
  return make_float2(mu, rsqrtf(var + eps));
}

// Running statistics of a wave's stored outputs, per channel: sums of (v - pivot), pivot = the wave's first stored value of the
// channel (values of one channel lie within a few standard deviations of each other: no cancellation in sum2 - sum^2 / n).
struct Stats {
  float piv[CH], s1[CH], s2[CH];
  float cnt;
  bool have;
};
__device__ __forceinline__ void stats_init(Stats &st) {
#pragma unroll
  for (int j = 0; j < CH; ++j) st.piv[j] = st.s1[j] = st.s2[j] = 0.f;
  st.cnt = 0.f;
  st.have = false;
}
__device__ __forceinline__ void stats_add(Stats &st, const float (&v)[CH], bool owner, int lane) {
  if (!st.have) {   // wave-uniform: lane 1 owns a position in a wave's first pass (the launcher never starts a pass beyond the chunk)
#pragma unroll
    for (int j = 0; j < CH; ++j) st.piv[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[j]), 1));
    st.have = true;
  }
  if (owner) {
    st.cnt += 1.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float d = v[j] - st.piv[j];
      st.s1[j] += d;
      st.s2[j] = fmaf(d, d, st.s2[j]);
    }
  }
}
// wave totals -> LDS part[wave][ch] = (n, mean, M2); then the first wave folds the waves in order and writes the chunk partial
template <bool FAST>
__device__ __forceinline__ void stats_finish(const Stats &st, float (*part)[CH][3], int wave, int nwaves, int lane, float *out, int G) {
  const float n = wave_sum_dpp(st.cnt);
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const float s1 = wave_sum_dpp(st.s1[j]), s2 = wave_sum_dpp(st.s2[j]);
    if (lane == 0) {
      const float m = n > 0.f ? div_t<FAST>(s1, n) : 0.f;
      part[wave][j][0] = n;
      part[wave][j][1] = st.piv[j] + m;
      part[wave][j][2] = fmaxf(s2 - s1 * m, 0.f);
    }
  }
  __syncthreads();
  if (wave == 0 && lane < CH && lane < G) {
    float n2 = 0.f, mean = 0.f, m2 = 0.f;
    for (int w = 0; w < nwaves; ++w) welford_merge_t<FAST>(n2, mean, m2, part[w][lane][0], part[w][lane][1], part[w][lane][2]);
    out[lane * 2] = mean;
    out[lane * 2 + 1] = m2;
  }
}

// The weights are wave-uniform: they are read with SCALAR loads (constant address space: s_load through the scalar cache) straight into
// SGPR operands of the FMAs -- no vector register, no LDS traffic.  All 272 of a tail pass cannot be live at once (102 SGPRs), so the
// pointer is passed through an empty asm that also consumes the previous accumulator: the loads of output channel n+1 cannot be
// scheduled before channel n-1 is finished (one channel of look-ahead), and cannot be hoisted out of the pass loop.
typedef const __attribute__((address_space(4))) float *cfloat_p;
__device__ __forceinline__ cfloat_p as_const(const float *p) { return (cfloat_p)(uintptr_t)p; }
__device__ __forceinline__ cfloat_p after(cfloat_p p, float dep) {
  asm volatile("" : "+s"(p) : "v"(dep));
  return p;
}

// lane l receives the value of lane l - 1 / l + 1 (DPP wave shifts: VALU only, no LDS crossbar); lane 0 / 63 receive 0
__device__ __forceinline__ float from_prev(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float from_next(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }

// o[n] = sum_{tap, c} w[n][tap * 8 + c] * (pv | cu | nx)[c]   (w: fp32 [8][24])
__device__ __forceinline__ void conv3_8(cfloat_p w, const float (&pv)[CH], const float (&cu)[CH], const float (&nx)[CH], float (&o)[CH]) {
  float dep = cu[0];
#pragma unroll
  for (int n = 0; n < CH; ++n) {
    const cfloat_p wr = after(w, dep) + n * 24;
    float a0 = 0.f, a1 = 0.f;   // two chains: packed FMAs
#pragma unroll
    for (int c = 0; c < CH; c += 2) {
      a0 = fmaf(wr[c], pv[c], a0);
      a1 = fmaf(wr[c + 1], pv[c + 1], a1);
      a0 = fmaf(wr[8 + c], cu[c], a0);
      a1 = fmaf(wr[8 + c + 1], cu[c + 1], a1);
      a0 = fmaf(wr[16 + c], nx[c], a0);
      a1 = fmaf(wr[16 + c + 1], nx[c + 1], a1);
    }
    o[n] = a0 + a1;
    if (n >= 1) dep = o[n - 1];
  }
}

// MAXT: the launch bound the kernel is compiled for -- 256 for the default four waves (the compiler may then use up to 256 VGPRs per
// lane: with a 1024-thread bound it is capped at 128, which cost 2-12 SGPR spills on every instantiation and scratch on d0_tail<float,2>),
// 1024 for the SF_D0_WAVES tuning hook.
template <typename T, int MAXT>
__global__ __launch_bounds__(MAXT) void d0_conv_kernel(const ConvThinArgs a) {
  constexpr bool FAST = sizeof(T) == 2;
  __shared__ __attribute__((aligned(16))) float prm[5][CH];        // GroupNorm scale, shift | bias | per-clip scale | per-clip add
  __shared__ float part[16][CH][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.nchw, ch = blockIdx.x - b * a.nchw;
  const int r0 = ch * a.rw, rows = min(a.rw, a.L - r0);
  const T *src = static_cast<const T *>(a.src);
  const T *res = static_cast<const T *>(a.res);
  T *out = static_cast<T *>(a.out);
  const cfloat_p w = as_const(a.w32);

  const size_t clip = (size_t)b * a.L;
  const bool res_g = res && !a.res_self;
  const int end = r0 + rows, stride = (blockDim.x >> 6) * SPAN;
  // one pass of look-ahead: the rows of pass i + 1 are in flight while pass i computes
  Raw8<T> xq, rq;
  rq.zero();
  auto fetch = [&](int base) {
    const int p = base + lane - 1;
    const bool in = p >= 0 && p < a.L, own = lane >= 1 && lane <= SPAN && p < end;
    xq.load(src + (clip + (in ? p : 0)) * a.src_ld);
    if (res_g) rq.load(res + (clip + (own ? p : r0)) * a.res_ld);
  };
  if (r0 + wave * SPAN < end) fetch(r0 + wave * SPAN);

  if (tid < CH) {
    prm[2][tid] = a.bias ? a.bias[tid] : 0.f;
    prm[3][tid] = a.bscale ? a.bscale[(size_t)b * a.bscale_ld + tid] : 1.f;
    prm[4][tid] = a.badd ? a.badd[(size_t)b * a.badd_ld + tid] : 0.f;
  }
  {   // GroupNorm statistics: half-wave hw handles channel hw (8 half-waves, 8 channels)
    const int hw = tid >> 5, l32 = tid & 31;
    if (hw >= CH) {
    } else if (a.pro == 1) {
      const float2 st = merge_partials<FAST>(a.stats_in + ((size_t)b * a.nch_in * a.G + hw) * 2, a.G, a.nch_in, a.chunk_in, a.L, a.eps, l32);
      if (l32 == 0) {
        const float s = st.y * a.gamma[hw];
        prm[0][hw] = s;
        prm[1][hw] = a.beta[hw] - st.x * s;
      }
    } else if (l32 == 0) {
      prm[0][hw] = 1.f;
      prm[1][hw] = 0.f;
    }
  }
  __syncthreads();
  float sc[CH], sh[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    sc[j] = prm[0][j];
    sh[j] = prm[1][j];
  }
  Stats st;
  stats_init(st);
  for (int base = r0 + wave * SPAN; base < end; base += stride) {
    const int p = base + lane - 1;
    const bool in = p >= 0 && p < a.L;
    const bool owner = lane >= 1 && lane <= SPAN && p < end;
    float x[CH], act[CH], pv[CH], nx[CH], o[CH], rv[CH];
    xq.get(x);
    rq.get(rv);
    if (base + stride < end) fetch(base + stride);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float y = fmaf(x[j], sc[j], sh[j]);
      if (a.pro == 1) y = silu_t<FAST>(y);
      act[j] = y;   // kept in fp32 (the MFMA kernels round their operand to the 16-bit type: the vector path has no reason to)
    }
    if (base <= 0 || base + SPAN >= a.L) {   // wave-uniform: only a wave at a clip edge has lanes outside, and the convolution pads the
#pragma unroll                              // ACTIVATED tensor with zeros
      for (int j = 0; j < CH; ++j) act[j] = in ? act[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      pv[j] = from_prev(act[j]);
      nx[j] = from_next(act[j]);
    }
    conv3_8(w, pv, act, nx, o);
    if (a.res_self) {
#pragma unroll
      for (int j = 0; j < CH; ++j) rv[j] = act[j];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) o[j] = o[j] + prm[2][j] + rv[j];
    if (a.bscale) {   // SkipModulate-style per-clip scale: not on the item convolutions
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] = (o[j] - rv[j]) * prm[3][j] + rv[j];
    }
    if (a.badd) {
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] += prm[4][j];
    }
    round8<T>(o);
    if (owner) store8<T>(out + (clip + p) * a.out_ld, o);
    if (a.stats_out) stats_add(st, o, owner, lane);
  }
  if (a.stats_out) stats_finish<FAST>(st, part, wave, blockDim.x >> 6, lane, a.stats_out + (((size_t)b * a.nchw + ch) * a.G) * 2, a.G);
}

// C2R: the REAL context channels (the context rows are padded to 8; the fp32 1x1 weights are not: rows of 8 + C2R)
template <typename T, int C2R, int MAXT>
__global__ __launch_bounds__(MAXT) void d0_tail_kernel(const ThinTailArgs a) {
  constexpr bool FAST = sizeof(T) == 2;
  __shared__ __attribute__((aligned(16))) float prm[6][CH];        // GroupNorm scale, shift | bias2 | 1 + modulation scale | modulation shift | bias3 + per-clip add
  __shared__ float part[16][CH][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.nchw, ch = blockIdx.x - b * a.nchw;
  const int r0 = ch * a.rw, rows = min(a.rw, a.L - r0);
  const T *hsrc = static_cast<const T *>(a.h), *xsrc = static_cast<const T *>(a.x), *csrc = static_cast<const T *>(a.ctx);
  T *out = static_cast<T *>(a.out);
  const cfloat_p w2 = as_const(a.w2_32), w3 = as_const(a.w3_32);
  constexpr int k3 = CH + C2R;
  const size_t clip = (size_t)b * a.L;
  const int end = r0 + rows, stride = (blockDim.x >> 6) * SPAN;
  Raw8<T> hq, xq, cq;
  auto fetch = [&](int base) {
    const int p = base + lane - 1;
    const size_t row = clip + ((p >= 0 && p < a.L) ? p : 0);
    hq.load(hsrc + row * CH);
    xq.load(xsrc + row * CH);
    cq.load(csrc + row * a.ctx_ld);
  };
  if (r0 + wave * SPAN < end) fetch(r0 + wave * SPAN);


  if (tid < CH) {
    prm[2][tid] = a.bias2[tid];
    prm[3][tid] = 1.0f + a.ss[(size_t)b * a.ss_ld + tid];
    prm[4][tid] = a.ss[(size_t)b * a.ss_ld + CH + tid];
    prm[5][tid] = a.bias3[tid] + (a.badd ? a.badd[(size_t)b * a.badd_ld + tid] : 0.f);
  }
  {
    const int hw = min(tid >> 5, CH - 1), l32 = tid & 31;
    const float2 st = merge_partials<FAST>(a.stats_in + ((size_t)b * a.nch_in * a.G + hw) * 2, a.G, a.nch_in, a.chunk_in, a.L, a.eps_gn, l32);
    if (l32 == 0 && (tid >> 5) < CH) {
      const float s = st.y * a.gamma[hw];
      prm[0][hw] = s;
      prm[1][hw] = a.beta[hw] - st.x * s;
    }
  }
  __syncthreads();
  float sc[CH], sh[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    sc[j] = prm[0][j];
    sh[j] = prm[1][j];
  }
  Stats st;
  stats_init(st);
  for (int base = r0 + wave * SPAN; base < end; base += stride) {
    const int p = base + lane - 1;
    const bool in = p >= 0 && p < a.L;
    const bool owner = lane >= 1 && lane <= SPAN && p < end;
    float hv[CH], xv[CH], cv[CH], act[CH], pv[CH], nx[CH], y[CH];
    hq.get(hv);
    xq.get(xv);
    cq.get(cv);
    if (base + stride < end) fetch(base + stride);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float v = silu_t<FAST>(fmaf(hv[j], sc[j], sh[j]));
      act[j] = v;
    }
    if (base <= 0 || base + SPAN >= a.L) {   // wave-uniform: a wave at a clip edge (zero padding of the activated tensor)
#pragma unroll
      for (int j = 0; j < CH; ++j) act[j] = in ? act[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      pv[j] = from_prev(act[j]);
      nx[j] = from_next(act[j]);
    }
    conv3_8(w2, pv, act, nx, y);
    // y = conv2 + bias + x; LayerNorm over the 8 channels of the position; modulate
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      y[j] += prm[2][j] + xv[j];   // (y and m stay in fp32: the unfused path rounds them to the 16-bit type because it stores them)
      sum += y[j];
    }
    const float mean = sum * 0.125f;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float d = y[j] - mean;
      sq = fmaf(d, d, sq);
    }
    const float rstd = rsqrtf(sq * 0.125f + a.eps_ln);
    float m[CH], z[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      m[j] = fmaf((y[j] - mean) * rstd, prm[3][j], prm[4][j]);
    }
    float dep = m[0];
#pragma unroll
    for (int n = 0; n < CH; ++n) {
      const cfloat_p wr = after(w3, dep) + n * k3;
      float a0 = m[n] + prm[5][n], a1 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; c += 2) {
        a0 = fmaf(wr[c], m[c], a0);
        a1 = fmaf(wr[c + 1], m[c + 1], a1);
      }
#pragma unroll
      for (int c = 0; c < C2R; ++c) a1 = fmaf(wr[CH + c], cv[c], a1);
      z[n] = a0 + a1;
      if (n >= 1) dep = z[n - 1];
    }
    round8<T>(z);
    if (owner) store8<T>(out + (clip + p) * CH, z);
    if (a.stats_out) stats_add(st, z, owner, lane);
  }
  if (a.stats_out) stats_finish<FAST>(st, part, wave, blockDim.x >> 6, lane, a.stats_out + (((size_t)b * a.nchw + ch) * a.G) * 2, a.G);
}

// waves per workgroup: tuning hook SF_D0_WAVES (4 ... 16)
int d0_threads() {
  static const int t = [] {
    const char *e = tune_env("SF_D0_WAVES");
    int v = e ? atoi(e) : 4;
    if (v < 4) v = 4;
    if (v > 16) v = 16;
    return v * 64;
  }();
  return t;
}

}  // namespace

// The vector kernels take over from B * L >= 256 K positions per launch (tuning hook SF_D0_MIN_ROWS; SF_NO_D0=1: never).  Below
// that a launch is one pass per wave and latency-bound either way, and the 16-wave MFMA workgroups finish sooner: 2 evaluations
// per step measured -0.7 % with these kernels, 32-64 evaluations +0.3 ... +1.5 % (profiles/r3_j_ab_d0.txt).
bool d0_enabled(int B, int L) {
  static const long min_rows = [] {
    if (tune_env("SF_NO_D0")) return -1L;
    const char *e = tune_env("SF_D0_MIN_ROWS");
    return e ? atol(e) : 262144L;
  }();
  return min_rows >= 0 && (long)B * L >= min_rows;
}

bool d0_conv_supported(const ConvThinArgs &a) {
  if (!d0_enabled(a.B, a.L)) return false;
  if (a.C != CH || a.N != CH || a.taps != 3 || a.C2 != 0 || a.up_shift != 0 || a.Ls != a.L) return false;
  if (a.pro != 0 && a.pro != 1) return false;
  if (a.G != CH || a.src_ld < CH || a.out_ld < CH || (a.src_ld % CH) || (a.out_ld % CH)) return false;
  if (a.res && !a.res_self && (a.res_ld % CH)) return false;
  if (a.pro == 1 && (!a.stats_in || !a.gamma || !a.beta)) return false;
  return a.w32 && a.rw >= 1 && a.nchw >= 1;
}
bool d0_tail_supported(const ThinTailArgs &a) {
  if (!d0_enabled(a.B, a.L)) return false;
  if (!a.w2_32 || !a.w3_32 || (a.c2real != 1 && a.c2real != 2 && a.c2real != 4 && a.c2real != 8)) return false;
  return a.C == CH && a.C2 == CH && a.G == CH && a.ctx_ld >= CH && (a.ctx_ld % CH) == 0 && a.ss && a.stats_in && a.bias2 && a.bias3 && a.rw >= 1;
}
hipError_t launch_d0_conv(int dt, const ConvThinArgs &a, hipStream_t s) {
  if (d0_threads() <= 256) SF_DISPATCH_STMT(dt, hipLaunchKernelGGL((d0_conv_kernel<T, 256>), dim3(a.B * a.nchw), dim3(d0_threads()), 0, s, a));
  else SF_DISPATCH_STMT(dt, hipLaunchKernelGGL((d0_conv_kernel<T, 1024>), dim3(a.B * a.nchw), dim3(d0_threads()), 0, s, a));
  return hipGetLastError();
}
hipError_t launch_d0_tail(int dt, const ThinTailArgs &a, hipStream_t s) {
#define SF_D0_TAIL(C2R)                                                                                                                  \
  do {                                                                                                                                   \
    if (d0_threads() <= 256) SF_DISPATCH_STMT(dt, hipLaunchKernelGGL((d0_tail_kernel<T, C2R, 256>), dim3(a.B * a.nchw), dim3(d0_threads()), 0, s, a)); \
    else SF_DISPATCH_STMT(dt, hipLaunchKernelGGL((d0_tail_kernel<T, C2R, 1024>), dim3(a.B * a.nchw), dim3(d0_threads()), 0, s, a));      \
  } while (0)
  switch (a.c2real) {
    case 1: SF_D0_TAIL(1); break;
    case 2: SF_D0_TAIL(2); break;
    case 4: SF_D0_TAIL(4); break;
    case 8: SF_D0_TAIL(8); break;
    default: return hipErrorInvalidValue;
  }
#undef SF_D0_TAIL
  return hipGetLastError();
}

}  // namespace sf
