// Backward kernels of the ResnetItem building block  y = Conv1d(SiLU(GroupNorm(x)))  and of plain Conv1d (the 1x1 InjectChannels
// conv), fp32, channels-last -- the first slice of the training step (the reference trains in fp32:
// exp/train_diffusion_gh.yaml:87 `precision: 32`, main/module_diffusion.py:73-82).
//
//   forward   u = xhat * gamma + beta,  xhat = (x - mean_g) * rstd_g ;  a = u * sigmoid(u) ;  y[b,l,n] = sum_{t,c} W[n][c][t] a[b, l+t-pad, c] + bias[n]
//   dgrad     da[b,l,c] = sum_{t,n} dy[b, l-t+pad, n] W[n][c][t]      -> the FORWARD implicit-GEMM kernels on flipped / transposed weights
//   wgrad     dW[n][c][t] = sum_{b,l} dy[b,l,n] a[b, l+t-pad, c]       -> conv_wgrad_kernel: a "TN" GEMM whose reduction runs over the rows;
//             fp32 MFMA 32x32x2 takes both operands with the reduction index across the two half-waves, so dy and a are read
//             straight from their row-major layout (32 consecutive floats per half-wave); split over row ranges, summed in a fixed order
//   db[n]     = sum_{b,l} dy[b,l,n]                                      -> col_sums_kernel (two deterministic stages)
//   GroupNorm+SiLU backward (gn_silu_bwd_kernel, one workgroup per (clip, group)):
//             du = da * sigmoid(u) (1 + u (1 - sigmoid(u)));  g = du * gamma;  dx = rstd (g - mean(g) - xhat mean(g xhat));
//             dgamma[c] = sum du xhat,  dbeta[c] = sum du   (per-clip partials, reduced over clips afterwards)
// No atomics anywhere: gradients are bit-reproducible.
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

// Conv1d weight (N, C, taps) -> dgrad weight matrix [c][t' * ldn + n] = W[n][c][taps-1-t']  (rows padded to ldn columns per tap)
__global__ void pack_dgrad_kernel(const float *__restrict__ w, int N, int C, int taps, int ldn, float *__restrict__ out) {
  const int64_t total = (int64_t)C * taps * ldn;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % ldn);
    const int64_t r = i / ldn;
    const int t = (int)(r % taps), c = (int)(r / taps);
    out[i] = n < N ? w[((int64_t)n * C + c) * taps + (taps - 1 - t)] : 0.f;
  }
}

// partial[s][n][q],  q = t * C + c:  sum over the rows of split s of dy[row][n] * a[row + t - pad][c]
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ act, int rows, int L, int C, int N,
                                                         int taps, int pad, int rows_per_split, float *__restrict__ partial) {
  __shared__ float red[4][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int Q = taps * C;
  const int n = blockIdx.x * 32 + fr;          // A operand: lane -> output channel
  const int q = blockIdx.y * 32 + fr;          // B operand: lane -> (tap, input channel)
  const int t = q < Q ? q / C : 0, c = q < Q ? q - t * C : 0;
  const int shift = t - pad;
  const int r_begin = blockIdx.z * rows_per_split, r_end = min(rows, r_begin + rows_per_split);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // wave w takes the row pairs w, w + 4, ...; inside a pair the half-wave picks the row (k index of the 32x32x2 MFMA)
  for (int r0 = r_begin + 2 * wave; r0 < r_end; r0 += 8) {
    const int r = r0 + fh;
    const bool rv = r < r_end;
    const float av = (rv && n < N) ? dy[(size_t)r * N + n] : 0.f;
    const int l = r % L;
    const int ls = l + shift;
    const bool bv = rv && q < Q && ls >= 0 && ls < L;
    const float bvv = bv ? act[(size_t)(r + shift) * C + c] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) red[wave][(i & 3) + 8 * (i >> 2) + 4 * fh][fr] = acc[i];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    const float v = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
    const int nn = blockIdx.x * 32 + i, qq = blockIdx.y * 32 + j;
    if (nn < N && qq < Q) partial[((size_t)blockIdx.z * N + nn) * Q + qq] = v;
  }
}

// dw[n][c][t] (PyTorch layout) = sum_s partial[s][n][t * C + c]
__global__ void wgrad_reduce_kernel(const float *__restrict__ partial, int S, int N, int C, int taps, float *__restrict__ dw) {
  const int64_t total = (int64_t)N * C * taps;
  const int Q = taps * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % taps);
    const int64_t r = i / taps;
    const int c = (int)(r % C), n = (int)(r / C);
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += partial[((size_t)k * N + n) * Q + (size_t)t * C + c];
    dw[i] = s;
  }
}

// part[blockIdx.y][col] = sum of x[row][col] over the rows of the slice (one thread per column, rows strided by gridDim.y)
__global__ void col_sums_kernel(const float *__restrict__ x, int64_t rows, int cols, int64_t rows_per_slice, float *__restrict__ part) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += x[r * cols + col];
  part[(size_t)blockIdx.y * cols + col] = s;
}
// out[j] = sum_k part[k][j]   (optionally two interleaved quantities: out2[j] from part2)
__global__ void slices_reduce_kernel(const float *__restrict__ part, int S, int cols, float *__restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[(size_t)k * cols + col];
  out[col] = s;
}

__device__ __forceinline__ float block_sum_256(float v, float *sh) {   // every thread gets the total; fixed order
  v = wave_sum_dpp(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One workgroup per (clip, group).  cpg must divide 256 (a thread then always meets the same channel).
__global__ __launch_bounds__(256) void gn_silu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ da, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, int L, int C, int G, float eps, float *__restrict__ dx,
                                                          float *__restrict__ dgb_part /* [B][2][C] */) {
  __shared__ float sh[4];
  __shared__ float chs[2][256];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int cpg = C / G;
  const int n_el = L * cpg;
  const float inv_n = 1.0f / (float)n_el;
  const float *xb = x + (size_t)b * L * C + g * cpg;
  const float *db = da + (size_t)b * L * C + g * cpg;
  float *ob = dx + (size_t)b * L * C + g * cpg;
  const int tc = threadIdx.x % cpg;   // this thread's channel inside the group (256 % cpg == 0)
  float s = 0.f;
  for (int i = threadIdx.x; i < n_el; i += 256) s += xb[(size_t)(i / cpg) * C + (i % cpg)];
  const float mean = block_sum_256(s, sh) * inv_n;
  float sq = 0.f;
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const float d = xb[(size_t)(i / cpg) * C + (i % cpg)] - mean;
    sq = fmaf(d, d, sq);
  }
  const float rstd = rsqrtf(block_sum_256(sq, sh) * inv_n + eps);
  const float gam = gamma[g * cpg + tc], bet = beta[g * cpg + tc];
  float s1 = 0.f, s2 = 0.f, dgam = 0.f, dbet = 0.f;
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const size_t off = (size_t)(i / cpg) * C + (i % cpg);
    const float xh = (xb[off] - mean) * rstd;
    const float u = fmaf(xh, gam, bet);
    const float sg = 1.0f / (1.0f + expf(-u));
    const float du = db[off] * sg * (1.0f + u * (1.0f - sg));
    const float g1 = du * gam;
    s1 += g1;
    s2 = fmaf(g1, xh, s2);
    dgam = fmaf(du, xh, dgam);
    dbet += du;
  }
  const float m1 = block_sum_256(s1, sh) * inv_n, m2 = block_sum_256(s2, sh) * inv_n;
  chs[0][threadIdx.x] = dgam;
  chs[1][threadIdx.x] = dbet;
  __syncthreads();
  if ((int)threadIdx.x < cpg) {   // fixed-order sum over the threads that share this channel
    float a0 = 0.f, a1 = 0.f;
    for (int k = threadIdx.x; k < 256; k += cpg) {
      a0 += chs[0][k];
      a1 += chs[1][k];
    }
    dgb_part[((size_t)b * 2 + 0) * C + g * cpg + threadIdx.x] = a0;
    dgb_part[((size_t)b * 2 + 1) * C + g * cpg + threadIdx.x] = a1;
  }
  for (int i = threadIdx.x; i < n_el; i += 256) {
    const size_t off = (size_t)(i / cpg) * C + (i % cpg);
    const float xh = (xb[off] - mean) * rstd;
    const float u = fmaf(xh, gam, bet);
    const float sg = 1.0f / (1.0f + expf(-u));
    const float g1 = db[off] * sg * (1.0f + u * (1.0f - sg)) * gam;
    ob[off] = rstd * (g1 - m1 - xh * m2);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// LayerNorm-modulate backward:  y = xhat (1 + s_b) + t_b,  xhat = LayerNorm_C(x; eps, no affine)
//   dx = rstd (g - mean_c(g) - xhat mean_c(g xhat)),  g = dy (1 + s_b);   ds_b[c] = sum_l dy xhat,  dt_b[c] = sum_l dy
// One workgroup per (clip, chunk of rows); a wave owns a row at a time, a lane the channels lane, lane + 64, ... (C <= 1024).
// The per-clip sums are accumulated per lane over the wave's rows, reduced over the four waves through LDS and written per chunk;
// slices_reduce_kernel adds the chunks (fixed order).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kLnMaxPer = 16;   // channels per lane: C <= 1024
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const float *__restrict__ x, const float *__restrict__ ss, int ss_ld, const float *__restrict__ dy,
                                                         int L, int C, float eps, int rows_per_chunk, float *__restrict__ dx,
                                                         float *__restrict__ dss_part /* [B][nchunk][2C] */) {
  __shared__ float red[4][2 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, chunk = blockIdx.y, nchunk = gridDim.y;
  const int per = (C + 63) / 64;
  float sc[kLnMaxPer], acs[kLnMaxPer], act[kLnMaxPer];
#pragma unroll
  for (int k = 0; k < kLnMaxPer; ++k) {
    const int c = lane + 64 * k;
    sc[k] = (k < per && c < C) ? 1.0f + (ss ? ss[(size_t)b * ss_ld + c] : 0.f) : 0.f;
    acs[k] = act[k] = 0.f;
  }
  const int l0 = chunk * rows_per_chunk, l1 = min(L, l0 + rows_per_chunk);
  const float inv_c = 1.0f / (float)C;
  for (int l = l0 + wave; l < l1; l += 4) {
    const size_t row = ((size_t)b * L + l) * C;
    float xv[kLnMaxPer], gv[kLnMaxPer];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxPer; ++k) {
      const int c = lane + 64 * k;
      xv[k] = (k < per && c < C) ? x[row + c] : 0.f;
      s += xv[k];
    }
    const float mean = wave_sum_dpp(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxPer; ++k) {
      const int c = lane + 64 * k;
      const float d = (k < per && c < C) ? xv[k] - mean : 0.f;
      q = fmaf(d, d, q);
    }
    const float rstd = rsqrtf(wave_sum_dpp(q) * inv_c + eps);
    float g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int k = 0; k < kLnMaxPer; ++k) {
      const int c = lane + 64 * k;
      const bool ok = k < per && c < C;
      const float xh = ok ? (xv[k] - mean) * rstd : 0.f;
      const float d = ok ? dy[row + c] : 0.f;
      xv[k] = xh;
      gv[k] = d * sc[k];
      g1 += gv[k];
      g2 = fmaf(gv[k], xh, g2);
      acs[k] = fmaf(d, xh, acs[k]);
      act[k] += d;
    }
    const float m1 = wave_sum_dpp(g1) * inv_c, m2 = wave_sum_dpp(g2) * inv_c;
#pragma unroll
    for (int k = 0; k < kLnMaxPer; ++k) {
      const int c = lane + 64 * k;
      if (k < per && c < C) dx[row + c] = rstd * (gv[k] - m1 - xv[k] * m2);
    }
  }
#pragma unroll
  for (int k = 0; k < kLnMaxPer; ++k) {
    const int c = lane + 64 * k;
    if (k < per && c < C) {
      red[wave][c] = acs[k];
      red[wave][1024 + c] = act[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float *o = dss_part + ((size_t)b * nchunk + chunk) * 2 * C;
    o[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    o[C + c] = (red[0][1024 + c] + red[1][1024 + c]) + (red[2][1024 + c] + red[3][1024 + c]);
  }
}
// out[b][j] = sum over chunks of part[b][chunk][j]
__global__ void chunks_reduce_kernel(const float *__restrict__ part, int nchunk, int cols, float *__restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (col >= cols) return;
  float s = 0.f;
  for (int k = 0; k < nchunk; ++k) s += part[((size_t)b * nchunk + k) * cols + col];
  out[(size_t)b * cols + col] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Multi-head softmax attention backward (head dim 64, fp32), two passes, no atomics:
//   pass Q (one workgroup per (clip, head, 16 queries)): S = q K^T scale, P = softmax(S), dP = dO V^T, D_i = sum_j P_ij dP_ij,
//            dS = P (dP - D) scale, dQ = dS K;  leaves lse_i = log sum_j exp(S_ij) and D_i for the second pass
//   pass K (one workgroup per (clip, head, 16 keys)): p_ij = exp(S_ij - lse_i); dV_j = sum_i p_ij dO_i; dK_j = sum_i p_ij (dP_ij - D_i) scale q_i
// q, dq, dout: (B, L, H*64) rows;  kv, dkv: (B, L, 2*H*64) rows (k | v).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int AD = 64, AQ = 16;
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ dout, int L, int H,
                                                         float scale, float *__restrict__ dq, float *__restrict__ lse, float *__restrict__ dsum) {
  extern __shared__ float sm[];
  float *qs = sm;                   // [AQ][AD]
  float *dos = qs + AQ * AD;        // [AQ][AD]
  float *P = dos + AQ * AD;         // [AQ][L]
  float *dP = P + (size_t)AQ * L;   // [AQ][L]
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * AQ, h = blockIdx.y, b = blockIdx.z;
  const int ldq = H * AD, ldkv = 2 * H * AD;
  const size_t rb = (size_t)b * L;
  for (int i = tid; i < AQ * AD; i += 256) {
    const int r = i / AD, d = i - r * AD;
    const bool ok = q0 + r < L;
    qs[i] = ok ? q[(rb + q0 + r) * ldq + h * AD + d] : 0.f;
    dos[i] = ok ? dout[(rb + q0 + r) * ldq + h * AD + d] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < AQ * L; e += 256) {   // S and dP: thread -> (query r, key j)
    const int r = e / L, j = e - r * L;
    const float *kp = kv + (rb + j) * ldkv + h * AD, *vp = kp + H * AD;
    float s = 0.f, dp = 0.f;
#pragma unroll 8
    for (int d = 0; d < AD; ++d) {
      s = fmaf(qs[r * AD + d], kp[d], s);
      dp = fmaf(dos[r * AD + d], vp[d], dp);
    }
    P[e] = s * scale;
    dP[e] = dp;
  }
  __syncthreads();
  {   // softmax statistics: 16 lanes per query row
    const int r = tid >> 4, sub = tid & 15;
    float mx = -INFINITY;
    for (int j = sub; j < L; j += 16) mx = fmaxf(mx, P[r * L + j]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    float sum = 0.f;
    for (int j = sub; j < L; j += 16) sum += expf(P[r * L + j] - mx);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    const float l = mx + logf(sum);
    float dd = 0.f;
    for (int j = sub; j < L; j += 16) {
      const float p = expf(P[r * L + j] - l);
      P[r * L + j] = p;
      dd = fmaf(p, dP[r * L + j], dd);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 16);
    for (int j = sub; j < L; j += 16) dP[r * L + j] = P[r * L + j] * (dP[r * L + j] - dd) * scale;   // dS
    if (sub == 0 && q0 + r < L) {
      lse[((size_t)b * H + h) * L + q0 + r] = l;
      dsum[((size_t)b * H + h) * L + q0 + r] = dd;
    }
  }
  __syncthreads();
  for (int e = tid; e < AQ * AD; e += 256) {   // dQ = dS K
    const int r = e / AD, d = e - r * AD;
    float acc = 0.f;
    for (int j = 0; j < L; ++j) acc = fmaf(dP[r * L + j], kv[(rb + j) * ldkv + h * AD + d], acc);
    if (q0 + r < L) dq[(rb + q0 + r) * ldq + h * AD + d] = acc;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_k_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ dout,
                                                         const float *__restrict__ lse, const float *__restrict__ dsum, int L, int H, float scale,
                                                         float *__restrict__ dkv) {
  __shared__ float ks[AQ][AD + 1], vs[AQ][AD + 1];
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * AQ, h = blockIdx.y, b = blockIdx.z;
  const int ldq = H * AD, ldkv = 2 * H * AD;
  const size_t rb = (size_t)b * L;
  for (int i = tid; i < AQ * AD; i += 256) {
    const int r = i / AD, d = i - r * AD;
    const bool ok = k0 + r < L;
    ks[r][d] = ok ? kv[(rb + k0 + r) * ldkv + h * AD + d] : 0.f;
    vs[r][d] = ok ? kv[(rb + k0 + r) * ldkv + (H + h) * AD + d] : 0.f;
  }
  __syncthreads();
  // thread -> (key j = tid / 16, dims 4 * (tid % 16) .. + 3); the 16 lanes of a key share its dot products
  const int j = tid >> 4, sub = tid & 15;
  float dk[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
  const float *lp = lse + ((size_t)b * H + h) * L, *dp_ = dsum + ((size_t)b * H + h) * L;
  for (int i = 0; i < L; ++i) {
    const float *qp = q + (rb + i) * ldq + h * AD + 4 * sub, *op = dout + (rb + i) * ldq + h * AD + 4 * sub;
    const f32x4 qv = *reinterpret_cast<const f32x4 *>(qp), ov = *reinterpret_cast<const f32x4 *>(op);
    float s = 0.f, dpv = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s = fmaf(qv[e], ks[j][4 * sub + e], s);
      dpv = fmaf(ov[e], vs[j][4 * sub + e], dpv);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 16);
      dpv += __shfl_xor(dpv, o, 16);
    }
    const float p = expf(s * scale - lp[i]);
    const float ds = p * (dpv - dp_[i]) * scale;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dv[e] = fmaf(p, ov[e], dv[e]);
      dk[e] = fmaf(ds, qv[e], dk[e]);
    }
  }
  if (k0 + j < L) {
    float *o = dkv + (rb + k0 + j) * ldkv + h * AD + 4 * sub;
    *reinterpret_cast<f32x4 *>(o) = f32x4{dk[0], dk[1], dk[2], dk[3]};
    *reinterpret_cast<f32x4 *>(o + H * AD) = f32x4{dv[0], dv[1], dv[2], dv[3]};
  }
}

}  // namespace

hipError_t launch_pack_dgrad(const float *w, int N, int C, int taps, int ldn, float *out, hipStream_t s) {
  const int64_t total = (int64_t)C * taps * ldn;
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid), dim3(256), 0, s, w, N, C, taps, ldn, out);
  return hipGetLastError();
}

int conv_wgrad_splits(int64_t rows, int N, int Q) {
  const int64_t tiles = (int64_t)((N + 31) / 32) * ((Q + 31) / 32);
  int64_t S = std::max<int64_t>(1, 2048 / std::max<int64_t>(tiles, 1));
  S = std::min<int64_t>(S, std::max<int64_t>(1, rows / 256));
  return (int)std::min<int64_t>(S, 1024);
}

hipError_t launch_conv_wgrad(const float *dy, const float *act, int B, int L, int C, int N, int taps, int pad, float *partial, int S, float *dw,
                             hipStream_t s) {
  const int rows = B * L, Q = taps * C;
  int rps = (rows + S - 1) / S;
  rps = (rps + 7) / 8 * 8;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3((N + 31) / 32, (Q + 31) / 32, S), dim3(256), 0, s, dy, act, rows, L, C, N, taps, pad, rps, partial);
  const int64_t total = (int64_t)N * Q;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, s, partial, S, N, C, taps, dw);
  return hipGetLastError();
}

hipError_t launch_col_sums(const float *x, int64_t rows, int cols, float *part, int S, float *out, hipStream_t s) {
  const int64_t rps = (rows + S - 1) / S;
  hipLaunchKernelGGL(col_sums_kernel, dim3((cols + 63) / 64, S), dim3(64), 0, s, x, rows, cols, rps, part);
  hipLaunchKernelGGL(slices_reduce_kernel, dim3((cols + 63) / 64), dim3(64), 0, s, part, S, cols, out);
  return hipGetLastError();
}

hipError_t launch_gn_silu_bwd(const float *x, const float *da, const float *gamma, const float *beta, int B, int L, int C, int G, float eps,
                              float *dx, float *dgb_part, float *dgb /* [2C] = dgamma | dbeta */, hipStream_t s) {
  const int cpg = C / G;
  if (G < 1 || C % G || cpg > 256 || (256 % cpg)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_silu_bwd_kernel, dim3(B * G), dim3(256), 0, s, x, da, gamma, beta, L, C, G, eps, dx, dgb_part);
  // dgb_part is [B][2][C]: rows b, columns (2C) -> column sums give [dgamma | dbeta]
  hipLaunchKernelGGL(slices_reduce_kernel, dim3((2 * C + 63) / 64), dim3(64), 0, s, dgb_part, B, 2 * C, dgb);
  return hipGetLastError();
}

int ln_mod_bwd_chunks(int L) { return std::max(1, std::min(64, (L + 63) / 64)); }

hipError_t launch_ln_modulate_bwd(const float *x, const float *ss, const float *dy, float eps, int B, int L, int C, float *dx, float *dss_part,
                                  float *dss, hipStream_t s) {
  if (C < 1 || C > 64 * kLnMaxPer) return hipErrorInvalidValue;
  const int nchunk = ln_mod_bwd_chunks(L);
  const int rpc = (L + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(ln_mod_bwd_kernel, dim3(B, nchunk), dim3(256), 0, s, x, ss, 2 * C, dy, L, C, eps, rpc, dx, dss_part);
  if (dss) hipLaunchKernelGGL(chunks_reduce_kernel, dim3((2 * C + 63) / 64, B), dim3(64), 0, s, dss_part, nchunk, 2 * C, dss);
  return hipGetLastError();
}

hipError_t launch_attention_bwd(const float *q, const float *kv, const float *dout, int B, int L, int H, int D, float *dq, float *dkv, float *lse,
                                float *dsum, hipStream_t s) {
  if (D != AD || L < 1) return hipErrorInvalidValue;
  const size_t lds = ((size_t)2 * AQ * AD + (size_t)2 * AQ * L) * sizeof(float);
  if (lds > 150 * 1024) return hipErrorInvalidValue;   // L <= ~1100: longer sequences need a tiled first pass
  static bool en = false;
  if (!en) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    en = true;
  }
  const float scale = 1.0f / sqrtf((float)AD);
  dim3 grid((L + AQ - 1) / AQ, H, B);
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), lds, s, q, kv, dout, L, H, scale, dq, lse, dsum);
  hipLaunchKernelGGL(attn_bwd_k_kernel, grid, dim3(256), 0, s, q, kv, dout, lse, dsum, L, H, scale, dkv);
  return hipGetLastError();
}

}  // namespace sf
