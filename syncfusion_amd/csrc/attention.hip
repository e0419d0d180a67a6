// Multi-head softmax attention over packed projections (a-unet AttentionBase):
//   out[b, i, h*D:(h+1)*D] = softmax_j( q_i . k_j / sqrt(D) ) v_j      D = 64, fp32 softmax.
//
// Flash-style: one workgroup per (clip, head, 64-query tile) streams 64-key K/V tiles through LDS with
// an online (running max / running sum) softmax, so the L x L score matrix is never materialised.
// This version accumulates on the vector ALUs in fp32 for both storage types (exact reference for
// the parity path); sequence lengths on this path are 44..352 (2048 at the reference's 2^18 length).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int D = 64;
constexpr int TQ = 64, TK = 64;
constexpr int LDK = 68;  // floats; 16-byte aligned rows, adjacent rows 4 banks apart
constexpr int LDP = 65;

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T *__restrict__ q, int ldq, const T *__restrict__ kv, int ldkv,
                                                        int L, int H, T *__restrict__ out, int ldo, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *Ks = reinterpret_cast<float *>(smem);
  float *Vs = Ks + TK * LDK;
  float *Ps = Vs + TK * LDK;
  constexpr int V = Vec16<T>::N;

  const int tid = threadIdx.x;
  const int r = tid >> 2, sub = tid & 3;
  const int q0 = blockIdx.x * TQ;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t rowbase = (size_t)b * L;
  const int qi = q0 + r;
  const bool qvalid = qi < L;

  float qreg[D];
  {
    const T *qp = q + (rowbase + (qvalid ? qi : 0)) * ldq + h * D;
#pragma unroll
    for (int c = 0; c < D / V; ++c) {
      Vec16<T> v = ld16<T>(qp + c * V);
#pragma unroll
      for (int j = 0; j < V; ++j) qreg[c * V + j] = v.get(j) * scale;
    }
  }

  float o[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) o[d] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int koff = h * D;          // k columns inside kv
  const int voff = H * D + h * D;  // v columns inside kv

  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();  // previous tile fully consumed
    {
      const int kr = tid >> 2;  // key row within the tile
      const int kj = k0 + kr;
      const bool kvalid = kj < L;
      const T *kp = kv + (rowbase + (kvalid ? kj : 0)) * ldkv;
#pragma unroll
      for (int c = 0; c < 16 / V; ++c) {
        const int d0 = sub * 16 + c * V;
        Vec16<T> kk = ld16<T>(kp + koff + d0);
        Vec16<T> vv = ld16<T>(kp + voff + d0);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          Ks[kr * LDK + d0 + j] = kvalid ? kk.get(j) : 0.f;
          Vs[kr * LDK + d0 + j] = kvalid ? vv.get(j) : 0.f;
        }
      }
    }
    __syncthreads();

    float s[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = sub + 4 * i;
      const float *kr = Ks + j * LDK;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < D / 4; ++c) {
        f32x4 kk = *reinterpret_cast<const f32x4 *>(kr + 4 * c);
        acc = fmaf(qreg[4 * c + 0], kk[0], acc);
        acc = fmaf(qreg[4 * c + 1], kk[1], acc);
        acc = fmaf(qreg[4 * c + 2], kk[2], acc);
        acc = fmaf(qreg[4 * c + 3], kk[3], acc);
      }
      s[i] = (k0 + j < L) ? acc : -INFINITY;
      tmax = fmaxf(tmax, s[i]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 1, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 2, 64));
    const float mnew = fmaxf(mrun, tmax);  // finite: every tile holds at least one valid key
    const float alpha = expf(mrun - mnew);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float p = expf(s[i] - mnew);
      psum += p;
      Ps[r * LDP + sub + 4 * i] = p;
    }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] *= alpha;
    __syncthreads();
    for (int j = 0; j < TK; ++j) {
      const float p = Ps[r * LDP + j];
      const float *vr = Vs + j * LDK + sub * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 vv = *reinterpret_cast<const f32x4 *>(vr + 4 * c);
        o[4 * c + 0] = fmaf(p, vv[0], o[4 * c + 0]);
        o[4 * c + 1] = fmaf(p, vv[1], o[4 * c + 1]);
        o[4 * c + 2] = fmaf(p, vv[2], o[4 * c + 2]);
        o[4 * c + 3] = fmaf(p, vv[3], o[4 * c + 3]);
      }
    }
  }
  lrun += __shfl_xor(lrun, 1, 64);
  lrun += __shfl_xor(lrun, 2, 64);
  if (!qvalid) return;
  const float inv = 1.0f / lrun;
  T *op = out + (rowbase + qi) * ldo + h * D + sub * 16;
#pragma unroll
  for (int c = 0; c < 16 / V; ++c) {
    Vec16<T> ov;
#pragma unroll
    for (int j = 0; j < V; ++j) ov.set(j, o[c * V + j] * inv);
    st16<T>(op + c * V, ov);
  }
}

template <typename T>
hipError_t go(const void *q, int ldq, const void *kv, int ldkv, int B, int L, int H, void *out, int ldo, hipStream_t s) {
  size_t lds = (size_t)(2 * TK * LDK + TQ * LDP) * sizeof(float);
  auto kern = attention_kernel<T>;
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  dim3 grid((L + TQ - 1) / TQ, H, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, static_cast<const T *>(q), ldq, static_cast<const T *>(kv), ldkv, L, H,
                     static_cast<T *>(out), ldo, 1.0f / sqrtf((float)D));
  return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_mfma(int dt, const void *q, int ldq, const void *kv, int ldkv, int B, int L, int H, int Dh, void *out, int ldo,
                                 hipStream_t s);

hipError_t launch_attention(int dt, const void *q, int ldq, const void *kv, int ldkv, int B, int L, int H, int Dh, void *out,
                            int ldo, hipStream_t s, bool x3, bool xfmt) {
  if (Dh != D || L <= 0) return hipErrorInvalidValue;
  if (xfmt && (dt != F32 || !x3 || !attention_f32_mfma_ok(ldq, ldkv, ldo, B, H))) return hipErrorInvalidValue;
  if (dt != F32 && (ldq % 8) == 0 && (ldkv % 8) == 0 && (ldo % 8) == 0)   // matrix-core path (attention_mfma.hip)
    return launch_attention_mfma(dt, q, ldq, kv, ldkv, B, L, H, Dh, out, ldo, s);
  if (dt == F32 && attention_f32_mfma_ok(ldq, ldkv, ldo, B, H) && !tune_env("SF_ATTN_F32_VALU"))   // fp32 matrix cores (attention_bwd.hip)
    return launch_attention_f32_mfma(static_cast<const float *>(q), ldq, static_cast<const float *>(kv), ldkv, B, L, H, static_cast<float *>(out), ldo, s,
                                     nullptr, x3, xfmt);
  return SF_DISPATCH_T(dt, go<T>(q, ldq, kv, ldkv, B, L, H, out, ldo, s));
}

}  // namespace sf
