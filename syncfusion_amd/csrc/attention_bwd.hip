// Multi-head softmax attention in fp32 (head dim 64) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): the forward of the fp32
// parity engine / training forward and the backward (the reference trains in fp32: exp/train_diffusion_gh.yaml:87), two backward
// kernels, no atomics -> gradients are reproducible bit for bit.
//
//   S = q k^T / 8,  P = softmax_j(S),  O = P v            (forward, attention.hip)
//   dP = dO v^T,  D_i = sum_j P_ij dP_ij = dO_i . O_i,  dS = P (dP - D) / 8
//   dQ = dS k,   dK = dS^T q,   dV = P^T dO
//
// Both kernels exploit one property of the 32x32x2 MFMA: a lane's 16 accumulator values lie in ONE column (lane % 32) and in the
// 16 rows r -> 8 (r / 4) + 4 (lane / 32) + r % 4.  A score tile computed with the reduction index of the NEXT product in its
// rows can therefore be fed straight back as the A operand of that product (lane -> output row, k-step r -> the row the
// accumulator register r belongs to); the B operand is read from LDS in the same row order.  No transposes through LDS, no
// shuffles in the inner loops.
//
//   Q pass (a wave owns 32 queries; loops over 32-key tiles staged in LDS):
//       S^T = K q^T  (lane -> query, registers -> keys)          pass 0: online log-sum-exp  -> lse_i
//       pass 1: S^T, dP^T = V dO^T, dS^T = P^T (dP^T - D) / 8;   dQ += dS K  (A = dS^T registers, B = K tile rows)
//   K pass (a wave owns 32 keys; loops over 32-query tiles staged in LDS):
//       S = Q k^T (lane -> key, registers -> queries), dP = dO v^T, P = exp(S / 8 - lse), dS as above
//       dV += P^T dO (A = P registers, B = dO tile rows),  dK += dS^T Q (A = dS registers, B = Q tile rows)
// Per 32 x 32 tile and wave: 96 (Q pass; + 32 in pass 0) and 128 (K pass) MFMAs of 64 cycles; the LDS reads are a few % of that.
#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int HD = 64, TILE = 32, PITCH = 66;   // row pitch 66 floats: the A-operand reads (32 rows, 2 adjacent columns) hit 64 distinct banks
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ f32x16 mfma32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

struct TileRegs {
  f32x4 v[2];
};

// rows [row0, row0 + 32) x 64 columns starting at col0 of a row-major (.., ld) matrix -> two float4 per thread (zeros past row L)
__device__ __forceinline__ void tile_fetch(TileRegs &t, const float *__restrict__ base, size_t rb, int row0, int L, int ld, int col0, int tid) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + 256 * e, r = idx >> 4, c4 = idx & 15;
    const int row = row0 + r;
    t.v[e] = row < L ? *reinterpret_cast<const f32x4 *>(base + (rb + row) * ld + col0 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

__device__ __forceinline__ void tile_store(float *lds, const TileRegs &t, int tid, int pitch = PITCH) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + 256 * e, r = idx >> 4, c4 = idx & 15;
    float *p = lds + r * pitch + 4 * c4;
    *reinterpret_cast<f32x2 *>(p) = f32x2{t.v[e][0], t.v[e][1]};
    *reinterpret_cast<f32x2 *>(p + 2) = f32x2{t.v[e][2], t.v[e][3]};
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ int acc_row(int r, int hf) { return 8 * (r >> 2) + 4 * hf + (r & 3); }

// Forward, same machinery (the fp32 parity engine and the training forward): S^T = K q^T puts a query in every lane, so the online
// softmax state (running max, running sum) is per lane and the output is accumulated TRANSPOSED, O^T = V^T P^T (A = V tile read
// column-wise from LDS, B = the probabilities in the registers): rescaling by exp(m_old - m_new) is a per-lane scalar.  The two
// half-waves of a query hold different keys; they agree on the tile maximum with one shuffle per tile.  64 MFMAs per 32 x 32 tile.
constexpr int VPITCH = 72;   // V tile rows are read 4 apart by the two half-waves: 4 * 72 = 32 (mod 64) banks apart
__global__ __launch_bounds__(256) void attn_fwd_f32_mfma_kernel(const float *__restrict__ q, int ldq, const float *__restrict__ kv, int ldkv, int L, int H,
                                                                float scale_log2e, float *__restrict__ out, int ldo, float *__restrict__ lse_out) {
  __shared__ __attribute__((aligned(16))) float Ks[TILE * PITCH];
  __shared__ __attribute__((aligned(16))) float Vs[TILE * VPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t rb = (size_t)b * L;
  const int qi = blockIdx.x * 128 + wave * 32 + li;
  const bool qv = qi < L;
  float qreg[32];
  {
    const size_t off = (rb + (qv ? qi : 0)) * ldq + h * HD + hf;
#pragma unroll
    for (int t = 0; t < 32; ++t) qreg[t] = qv ? q[off + 2 * t] * scale_log2e : 0.f;   // scores in log2 units: exp2f below
  }
  const int nkt = (L + TILE - 1) / TILE;
  float m = -INFINITY, lsum = 0.f;
  f32x16 o0 = zero16(), o1 = zero16();
  TileRegs pk, pv;
  tile_fetch(pk, kv, rb, 0, L, ldkv, h * HD, tid);
  tile_fetch(pv, kv, rb, 0, L, ldkv, (H + h) * HD, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_store(Ks, pk, tid);
    tile_store(Vs, pv, tid, VPITCH);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_fetch(pk, kv, rb, (kt + 1) * TILE, L, ldkv, h * HD, tid);
      tile_fetch(pv, kv, rb, (kt + 1) * TILE, L, ldkv, (H + h) * HD, tid);
    }
    f32x16 s = zero16();
#pragma unroll
    for (int t = 0; t < 32; ++t) s = mfma32x2(Ks[li * PITCH + 2 * t + hf], qreg[t], s);
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = (kt * TILE + acc_row(r, hf) < L) ? s[r] : -INFINITY;
      tm = fmaxf(tm, s[r]);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32));          // key kt * 32 is valid, so the combined maximum is finite
    const float mn = fmaxf(m, tm), alpha = exp2f(m - mn);
    float add = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = exp2f(s[r] - mn);
      add += s[r];
    }
    lsum = lsum * alpha + add;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] *= alpha;
      o1[r] *= alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *vrow = Vs + acc_row(r, hf) * VPITCH + li;
      o0 = mfma32x2(vrow[0], s[r], o0);
      o1 = mfma32x2(vrow[32], s[r], o1);
    }
  }
  lsum += __shfl_xor(lsum, 32);
  // log sum_j exp(S_ij / 8) in natural units for the backward pass (the running maximum is shared by the half-waves): saves its pass 0
  if (lse_out && hf == 0 && qv) lse_out[((size_t)b * H + h) * L + qi] = (m + log2f(lsum)) * 0.6931471805599453f;
  if (qv) {
    const float inv = 1.0f / lsum;
    float *op = out + (rb + qi) * ldo + h * HD + 4 * hf;   // O^T: registers 4 g .. 4 g + 3 = head dims 8 g + 4 hf .. + 3 of this lane's query
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<f32x4 *>(op + 8 * g) = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      *reinterpret_cast<f32x4 *>(op + 32 + 8 * g) = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
    }
  }
}

// The same forward on split fp16 operands (SF_F32X; common.h X3P<X3_F16>): q (pre-scaled, so that the scores come out in log2 units) is split
// once into registers; the K and V tiles are split while they are staged and sit in LDS as (hi, lo') fp16 images, K read row-wise (a key
// per lane, 8 consecutive head dims), V gathered column-wise by the hardware transpose read; the probabilities (in [0, 1]) are split in
// registers and fed back as the B operand in the accumulator's own key order (element j of half h = key 16 s + 8 (j / 4) + 4 h + j % 4:
// two blocks of four consecutive keys = two transposed reads of V per fragment).  24 MFMAs of 32 cycles per 32 x 32 tile instead of 64 of 64.
constexpr int KP16 = 72, VP16 = 96;   // fp16 row pitches: K rows 144 B (row reads conflict-free), V rows 192 B (rows 64 B apart modulo 256)
typedef short v4i16_a __attribute__((ext_vector_type(4)));
typedef short v8i16_a __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) v4i16_a lds_v4i16_a;
typedef f16 f16x4_a __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tile_store_x3(f16 *hi, f16 *lo, const TileRegs &t, int tid, int pitch) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + 256 * e, r = idx >> 4, c4 = idx & 15;
    f16x4_a h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f16 a, b;
      x3_split1<X3_F16>(t.v[e][j], a, b);
      h[j] = a;
      l[j] = b;
    }
    *reinterpret_cast<f16x4_a *>(hi + r * pitch + 4 * c4) = h;
    *reinterpret_cast<f16x4_a *>(lo + r * pitch + 4 * c4) = l;
  }
}

__global__ __launch_bounds__(256) void attn_fwd_x3_kernel(const float *__restrict__ q, int ldq, const float *__restrict__ kv, int ldkv, int L, int H,
                                                          float scale_log2e, float *__restrict__ out, int ldo, float *__restrict__ lse_out, const int xfmt) {
  __shared__ __attribute__((aligned(16))) f16 KsH[TILE * KP16], KsL[TILE * KP16];
  __shared__ __attribute__((aligned(16))) f16 VsH[TILE * VP16], VsL[TILE * VP16];
  using XP = X3P<X3_F16>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t rb = (size_t)b * L;
  const int qi = blockIdx.x * 128 + wave * 32 + li;
  const bool qv = qi < L;
  // B operand of S^T = K q^T: this lane's query, head dims 16 s + 8 hf .. + 7 of step s
  f16x8 qh[4], ql[4];
  {
    const float *qp = q + (rb + (qv ? qi : 0)) * ldq + h * HD + 8 * hf;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 a = qv ? *reinterpret_cast<const f32x4 *>(qp + 16 * s) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 c = qv ? *reinterpret_cast<const f32x4 *>(qp + 16 * s + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] *= scale_log2e;
        c[e] *= scale_log2e;
      }
      x3_split<X3_F16>(a, c, qh[s], ql[s]);
    }
  }
  const int nkt = (L + TILE - 1) / TILE;
  float m = -INFINITY, lsum = 0.f;
  f32x16 o0 = zero16(), o1 = zero16(), o0l = zero16(), o1l = zero16();
  TileRegs pk, pv;
  tile_fetch(pk, kv, rb, 0, L, ldkv, h * HD, tid);
  tile_fetch(pv, kv, rb, 0, L, ldkv, (H + h) * HD, tid);
  // transposed-read geometry (cdna_hip_programming.md T10): lane 4 q + p of a 16-lane group addresses key row q, head dims 4 p .. 4 p + 3
  const int tr_off = (4 * hf + ((lane & 15) >> 2)) * VP16 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  auto vfrag = [&](const f16 *base) {
    const v4i16_a a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_a *)(base));
    const v4i16_a c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_a *)(base + 8 * VP16));
    const v8i16_a both = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    return __builtin_bit_cast(f16x8, both);
  };
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_store_x3(KsH, KsL, pk, tid, KP16);
    tile_store_x3(VsH, VsL, pv, tid, VP16);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_fetch(pk, kv, rb, (kt + 1) * TILE, L, ldkv, h * HD, tid);
      tile_fetch(pv, kv, rb, (kt + 1) * TILE, L, ldkv, (H + h) * HD, tid);
    }
    f32x16 sm = zero16(), sl = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 kh = *reinterpret_cast<const f16x8 *>(KsH + li * KP16 + 16 * s + 8 * hf);
      const f16x8 kl = *reinterpret_cast<const f16x8 *>(KsL + li * KP16 + 16 * s + 8 * hf);
      x3_mfma<X3_F16>(kh, kl, qh[s], ql[s], sm, sl);
    }
    float sv[16];
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = fmaf(sl[r], XP::INV, sm[r]);
      sv[r] = (kt * TILE + acc_row(r, hf) < L) ? x : -INFINITY;
      tm = fmaxf(tm, sv[r]);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32));
    const float mn = fmaxf(m, tm), alpha = exp2f(m - mn);
    float add = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sv[r] = exp2f(sv[r] - mn);
      add += sv[r];
    }
    lsum = lsum * alpha + add;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0[r] *= alpha;
      o1[r] *= alpha;
      o0l[r] *= alpha;
      o1l[r] *= alpha;
    }
    // O^T += V^T P^T: step s covers the keys of accumulator registers 8 s .. 8 s + 7 (this lane's B operand), A = V columns from the
    // row-major images by two transposed reads (keys 16 s + 4 hf + 0..3 and + 8)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f16 a, c;
        x3_split1<X3_F16>(sv[8 * s + j], a, c);
        ph[j] = a;
        pl[j] = c;
      }
      const int o = (16 * s) * VP16 + tr_off;
      x3_mfma<X3_F16>(vfrag(VsH + o), vfrag(VsL + o), ph, pl, o0, o0l);
      x3_mfma<X3_F16>(vfrag(VsH + o + 32), vfrag(VsL + o + 32), ph, pl, o1, o1l);
    }
  }
  lsum += __shfl_xor(lsum, 32);
  if (lse_out && hf == 0 && qv) lse_out[((size_t)b * H + h) * L + qi] = (m + log2f(lsum)) * 0.6931471805599453f;
  if (qv) {
    const float inv = 1.0f / lsum;
    float *op = out + (rb + qi) * ldo + h * HD + 4 * hf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = fmaf(o0l[4 * g + e], XP::INV, o0[4 * g + e]) * inv;
        c[e] = fmaf(o1l[4 * g + e], XP::INV, o1[4 * g + e]) * inv;
      }
      if (xfmt) {   // rows pre-split for the output projection (common.h, st4_x3)
        st4_x3(out + (rb + qi) * ldo, h * HD + 4 * hf + 8 * g, a);
        st4_x3(out + (rb + qi) * ldo, h * HD + 4 * hf + 8 * g + 32, c);
      } else {
        *reinterpret_cast<f32x4 *>(op + 8 * g) = a;
        *reinterpret_cast<f32x4 *>(op + 32 + 8 * g) = c;
      }
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_q_mfma_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ o,
                                                              const float *__restrict__ dout, int L, int H, float scale, float *__restrict__ dq,
                                                              float *__restrict__ lse_out, float *__restrict__ dsum_out, const float *__restrict__ lse_in) {
  __shared__ __attribute__((aligned(16))) float Ks[TILE * PITCH];
  __shared__ __attribute__((aligned(16))) float Vs[TILE * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ldq = H * HD, ldkv = 2 * H * HD;
  const size_t rb = (size_t)b * L;
  const int qbase = blockIdx.x * 128 + wave * 32;
  const int qi = qbase + li;
  const bool qv = qi < L;
  // this lane's query row as B operands: element d = 2 t + hf of q_i and dO_i;  D_i = dO_i . O_i
  float qreg[32], doreg[32];
  float dd = 0.f;
  {
    const size_t off = (rb + (qv ? qi : 0)) * ldq + h * HD + hf;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      qreg[t] = qv ? q[off + 2 * t] : 0.f;
      doreg[t] = qv ? dout[off + 2 * t] : 0.f;
      dd = fmaf(doreg[t], qv ? o[off + 2 * t] : 0.f, dd);
    }
  }
  dd += __shfl_xor(dd, 32);
  const int nkt = (L + TILE - 1) / TILE;
  TileRegs pk, pv;
  // ---- pass 0: lse_i = log sum_j exp(S_ij) -- skipped when the forward pass kept it (lse_in) --------------------------------
  float m = -INFINITY, ssum = 0.f;
  if (!lse_in) tile_fetch(pk, kv, rb, 0, L, ldkv, h * HD, tid);
  for (int kt = 0; kt < (lse_in ? 0 : nkt); ++kt) {
    __syncthreads();
    tile_store(Ks, pk, tid);
    __syncthreads();
    if (kt + 1 < nkt) tile_fetch(pk, kv, rb, (kt + 1) * TILE, L, ldkv, h * HD, tid);
    f32x16 s = zero16();
#pragma unroll
    for (int t = 0; t < 32; ++t) s = mfma32x2(Ks[li * PITCH + 2 * t + hf], qreg[t], s);
    float sv[16];
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sv[r] = (kt * TILE + acc_row(r, hf) < L) ? s[r] * scale : -INFINITY;
      tm = fmaxf(tm, sv[r]);
    }
    const float mn = fmaxf(m, tm);
    if (mn > -INFINITY) {   // a half-wave can meet a tile without a valid key of its own (ragged last tile)
      float add = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) add += expf(sv[r] - mn);
      ssum = ssum * expf(m - mn) + add;
      m = mn;
    }
  }
  float lse;
  if (lse_in) {
    lse = qv ? lse_in[((size_t)b * H + h) * L + qi] : 0.f;
  } else {
    const float m2 = __shfl_xor(m, 32), s2 = __shfl_xor(ssum, 32);
    const float M = fmaxf(m, m2);
    const float tot = (m > -INFINITY ? ssum * expf(m - M) : 0.f) + (m2 > -INFINITY ? s2 * expf(m2 - M) : 0.f);
    lse = M + logf(tot);
  }
  if (hf == 0 && qv) {
    if (!lse_in) lse_out[((size_t)b * H + h) * L + qi] = lse;
    dsum_out[((size_t)b * H + h) * L + qi] = dd;
  }
  // ---- pass 1: dQ ------------------------------------------------------------------------------------------------------------
  f32x16 acc0 = zero16(), acc1 = zero16();
  tile_fetch(pk, kv, rb, 0, L, ldkv, h * HD, tid);
  tile_fetch(pv, kv, rb, 0, L, ldkv, (H + h) * HD, tid);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_store(Ks, pk, tid);
    tile_store(Vs, pv, tid);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_fetch(pk, kv, rb, (kt + 1) * TILE, L, ldkv, h * HD, tid);
      tile_fetch(pv, kv, rb, (kt + 1) * TILE, L, ldkv, (H + h) * HD, tid);
    }
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      s = mfma32x2(Ks[li * PITCH + 2 * t + hf], qreg[t], s);
      dp = mfma32x2(Vs[li * PITCH + 2 * t + hf], doreg[t], dp);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (kt * TILE + acc_row(r, hf) < L) ? expf(s[r] * scale - lse) : 0.f;
      ds[r] = p * (dp[r] - dd) * scale;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float *krow = Ks + acc_row(r, hf) * PITCH + li;
      acc0 = mfma32x2(ds[r], krow[0], acc0);
      acc1 = mfma32x2(ds[r], krow[32], acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = qbase + acc_row(r, hf);
    if (row < L) {
      float *p = dq + (rb + row) * ldq + h * HD + li;
      p[0] = acc0[r];
      p[32] = acc1[r];
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kv_mfma_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ dout,
                                                               const float *__restrict__ lse, const float *__restrict__ dsum, int L, int H, float scale,
                                                               float *__restrict__ dkv) {
  __shared__ __attribute__((aligned(16))) float Qs[TILE * PITCH];
  __shared__ __attribute__((aligned(16))) float Os[TILE * PITCH];
  __shared__ float lse_s[TILE], dsum_s[TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ldq = H * HD, ldkv = 2 * H * HD;
  const size_t rb = (size_t)b * L;
  const int kbase = blockIdx.x * 128 + wave * 32;
  const int kj = kbase + li;
  const bool kvld = kj < L;
  float kreg[32], vreg[32];
  {
    const size_t off = (rb + (kvld ? kj : 0)) * ldkv + h * HD + hf;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      kreg[t] = kvld ? kv[off + 2 * t] : 0.f;
      vreg[t] = kvld ? kv[off + H * HD + 2 * t] : 0.f;
    }
  }
  const float *lp = lse + ((size_t)b * H + h) * L, *dp_ = dsum + ((size_t)b * H + h) * L;
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
  const int nqt = (L + TILE - 1) / TILE;
  TileRegs pq, po;
  float pl = 0.f, pd = 0.f;
  tile_fetch(pq, q, rb, 0, L, ldq, h * HD, tid);
  tile_fetch(po, dout, rb, 0, L, ldq, h * HD, tid);
  if (tid < TILE) {
    pl = tid < L ? lp[tid] : INFINITY;     // +inf -> P = exp(-inf) = 0 for the rows past the end
    pd = tid < L ? dp_[tid] : 0.f;
  }
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    tile_store(Qs, pq, tid);
    tile_store(Os, po, tid);
    if (tid < TILE) {
      lse_s[tid] = pl;
      dsum_s[tid] = pd;
    }
    __syncthreads();
    if (qt + 1 < nqt) {
      tile_fetch(pq, q, rb, (qt + 1) * TILE, L, ldq, h * HD, tid);
      tile_fetch(po, dout, rb, (qt + 1) * TILE, L, ldq, h * HD, tid);
      if (tid < TILE) {
        const int i = (qt + 1) * TILE + tid;
        pl = i < L ? lp[i] : INFINITY;
        pd = i < L ? dp_[i] : 0.f;
      }
    }
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      s = mfma32x2(Qs[li * PITCH + 2 * t + hf], kreg[t], s);
      dp = mfma32x2(Os[li * PITCH + 2 * t + hf], vreg[t], dp);
    }
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = acc_row(r, hf);
      p[r] = expf(s[r] * scale - lse_s[i]);
      ds[r] = p[r] * (dp[r] - dsum_s[i]) * scale;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = acc_row(r, hf);
      const float *orow = Os + i * PITCH + li, *qrow = Qs + i * PITCH + li;
      dv0 = mfma32x2(p[r], orow[0], dv0);
      dv1 = mfma32x2(p[r], orow[32], dv1);
      dk0 = mfma32x2(ds[r], qrow[0], dk0);
      dk1 = mfma32x2(ds[r], qrow[32], dk1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = kbase + acc_row(r, hf);
    if (row < L) {
      float *pk_ = dkv + (rb + row) * ldkv + h * HD + li;
      pk_[0] = dk0[r];
      pk_[32] = dk1[r];
      pk_[H * HD] = dv0[r];
      pk_[H * HD + 32] = dv1[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The backward pass on split 16-bit operands (SF_F32X).  Same tiling and the same accumulator-as-operand trick as the fp32 kernels above;
// 48 MFMAs of 32 cycles per 32 x 32 tile and wave instead of 96 / 128 of 64.  Which split a product takes follows its operands:
//   S  = q k^T   : both are forward activations -> fp16 hi / 2048-scaled lo (22 bits per operand): P = exp(S / 8 - lse) amplifies the absolute
//                  error of S, so the scores get the accurate mode;
//   dP, dQ, dK, dV: one operand is a gradient (dO, dS), which spans the whole fp32 exponent range -> bf16 hi / lo (no range restriction).
// Tiles that are read row-wise (a row per lane, 8 consecutive head dims: the A operand of S and dP) sit in LDS at a 144-byte pitch, tiles
// that are read column-wise (the B operand of dQ / dK / dV: 8 consecutive rows of one head dim per lane) at a 192-byte pitch and are
// gathered by the hardware transpose read; a tile that is read both ways is staged twice.  Requires the forward pass's log-sum-exp.
// ---------------------------------------------------------------------------------------------------------------
typedef bf16 bf16x4_a __attribute__((ext_vector_type(4)));
template <int MODE> __device__ __forceinline__ void tile_store_split(typename X3P<MODE>::elem *hi, typename X3P<MODE>::elem *lo, const TileRegs &t, int tid,
                                                                    int pitch) {
  using E = typename X3P<MODE>::elem;
  typedef E E4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + 256 * e, r = idx >> 4, c4 = idx & 15;
    E4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      E a, b;
      x3_split1<MODE>(t.v[e][j], a, b);
      h[j] = a;
      l[j] = b;
    }
    *reinterpret_cast<E4 *>(hi + r * pitch + 4 * c4) = h;
    *reinterpret_cast<E4 *>(lo + r * pitch + 4 * c4) = l;
  }
}
// a lane's 64 values of one row (head dims 16 s + 8 hf .. + 7 of step s) as split B-operand fragments
template <int MODE>
__device__ __forceinline__ void row_frags(const float *p, bool valid, float mul, typename X3P<MODE>::v8 (&hi)[4], typename X3P<MODE>::v8 (&lo)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x4 a = valid ? *reinterpret_cast<const f32x4 *>(p + 16 * s) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 c = valid ? *reinterpret_cast<const f32x4 *>(p + 16 * s + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] *= mul;
      c[e] *= mul;
    }
    x3_split<MODE>(a, c, hi[s], lo[s]);
  }
}
template <int MODE> __device__ __forceinline__ void split8(const float *v, typename X3P<MODE>::v8 &hi, typename X3P<MODE>::v8 &lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    typename X3P<MODE>::elem a, b;
    x3_split1<MODE>(v[j], a, b);
    hi[j] = a;
    lo[j] = b;
  }
}
__device__ __forceinline__ bf16x8 tr_frag_bf16(const bf16 *base, int pitch) {
  const v4i16_a a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_a *)(base));
  const v4i16_a c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_a *)(base + 8 * pitch));
  const v8i16_a both = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
  return __builtin_bit_cast(bf16x8, both);
}

__global__ __launch_bounds__(256) void attn_bwd_q_x3_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ o,
                                                            const float *__restrict__ dout, int L, int H, float scale, float *__restrict__ dq,
                                                            float *__restrict__ dsum_out, const float *__restrict__ lse_in) {
  __shared__ __attribute__((aligned(16))) f16 KrH[TILE * KP16], KrL[TILE * KP16];       // K rows, fp16 split: S^T = K q^T
  __shared__ __attribute__((aligned(16))) bf16 VrH[TILE * KP16], VrL[TILE * KP16];      // V rows, bf16 split: dP^T = V dO^T
  __shared__ __attribute__((aligned(16))) bf16 KcH[TILE * VP16], KcL[TILE * VP16];      // K again, bf16 split, column reads: dQ += dS K
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ldq = H * HD, ldkv = 2 * H * HD;
  const size_t rb = (size_t)b * L;
  const int qbase = blockIdx.x * 128 + wave * 32;
  const int qi = qbase + li;
  const bool qv = qi < L;
  f16x8 qh[4], ql[4];
  bf16x8 doh[4], dol[4];
  float dd = 0.f;
  {
    const size_t off = (rb + (qv ? qi : 0)) * ldq + h * HD + 8 * hf;
    row_frags<X3_F16>(q + off, qv, scale, qh, ql);      // the scores come out already scaled by 1 / 8
    row_frags<X3_BF16>(dout + off, qv, 1.0f, doh, dol);
    if (qv) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) dd = fmaf(dout[off + 16 * s + e], o[off + 16 * s + e], dd);   // D_i = dO_i . O_i (this half's 32 head dims)
    }
  }
  dd += __shfl_xor(dd, 32);
  const float lse = qv ? lse_in[((size_t)b * H + h) * L + qi] : 0.f;
  if (hf == 0 && qv) dsum_out[((size_t)b * H + h) * L + qi] = dd;
  const int nkt = (L + TILE - 1) / TILE;
  TileRegs pk, pv;
  f32x16 acc0 = zero16(), acc1 = zero16();   // (bf16 products: one accumulator each, x3_mfma1_bf16)
  tile_fetch(pk, kv, rb, 0, L, ldkv, h * HD, tid);
  tile_fetch(pv, kv, rb, 0, L, ldkv, (H + h) * HD, tid);
  const int tr_off = (4 * hf + ((lane & 15) >> 2)) * VP16 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    tile_store_split<X3_F16>(KrH, KrL, pk, tid, KP16);
    tile_store_split<X3_BF16>(KcH, KcL, pk, tid, VP16);
    tile_store_split<X3_BF16>(VrH, VrL, pv, tid, KP16);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_fetch(pk, kv, rb, (kt + 1) * TILE, L, ldkv, h * HD, tid);
      tile_fetch(pv, kv, rb, (kt + 1) * TILE, L, ldkv, (H + h) * HD, tid);
    }
    f32x16 sm = zero16(), sl = zero16(), dpm = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int o16 = li * KP16 + 16 * s + 8 * hf;
      x3_mfma<X3_F16>(*reinterpret_cast<const f16x8 *>(KrH + o16), *reinterpret_cast<const f16x8 *>(KrL + o16), qh[s], ql[s], sm, sl);
      x3_mfma1_bf16(*reinterpret_cast<const bf16x8 *>(VrH + o16), *reinterpret_cast<const bf16x8 *>(VrL + o16), doh[s], dol[s], dpm);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sc = fmaf(sl[r], X3P<X3_F16>::INV, sm[r]);
      const float p = (kt * TILE + acc_row(r, hf) < L) ? expf(sc - lse) : 0.f;
      ds[r] = p * (dpm[r] - dd) * scale;
    }
    // dQ += dS K: the lane's dS values in accumulator order are its A operand (keys 16 s + 8 (j / 4) + 4 hf + j % 4), K columns by two
    // transposed reads per fragment
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 dh, dl;
      split8<X3_BF16>(ds + 8 * s, dh, dl);
      const int ob = (16 * s) * VP16 + tr_off;
      x3_mfma1_bf16(dh, dl, tr_frag_bf16(KcH + ob, VP16), tr_frag_bf16(KcL + ob, VP16), acc0);
      x3_mfma1_bf16(dh, dl, tr_frag_bf16(KcH + ob + 32, VP16), tr_frag_bf16(KcL + ob + 32, VP16), acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = qbase + acc_row(r, hf);
    if (row < L) {
      float *p = dq + (rb + row) * ldq + h * HD + li;
      p[0] = acc0[r];
      p[32] = acc1[r];
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_kv_x3_kernel(const float *__restrict__ q, const float *__restrict__ kv, const float *__restrict__ dout,
                                                             const float *__restrict__ lse, const float *__restrict__ dsum, int L, int H, float scale,
                                                             float *__restrict__ dkv) {
  __shared__ __attribute__((aligned(16))) f16 QrH[TILE * KP16], QrL[TILE * KP16];       // Q rows, fp16 split: S = Q k^T
  __shared__ __attribute__((aligned(16))) bf16 OrH[TILE * KP16], OrL[TILE * KP16];      // dO rows, bf16 split: dP = dO v^T
  __shared__ __attribute__((aligned(16))) bf16 QcH[TILE * VP16], QcL[TILE * VP16];      // Q again, bf16 split, column reads: dK += dS^T Q
  __shared__ __attribute__((aligned(16))) bf16 OcH[TILE * VP16], OcL[TILE * VP16];      // dO again, column reads: dV += P^T dO
  __shared__ float lse_s[TILE], dsum_s[TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hf = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ldq = H * HD, ldkv = 2 * H * HD;
  const size_t rb = (size_t)b * L;
  const int kbase = blockIdx.x * 128 + wave * 32;
  const int kj = kbase + li;
  const bool kvld = kj < L;
  f16x8 kh[4], kl[4];
  bf16x8 vh[4], vl[4];
  {
    const size_t off = (rb + (kvld ? kj : 0)) * ldkv + h * HD + 8 * hf;
    row_frags<X3_F16>(kv + off, kvld, scale, kh, kl);
    row_frags<X3_BF16>(kv + off + H * HD, kvld, 1.0f, vh, vl);
  }
  const float *lp = lse + ((size_t)b * H + h) * L, *dp_ = dsum + ((size_t)b * H + h) * L;
  f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();   // (bf16 products: one accumulator each)
  const int nqt = (L + TILE - 1) / TILE;
  TileRegs pq, po;
  float pl = 0.f, pd = 0.f;
  tile_fetch(pq, q, rb, 0, L, ldq, h * HD, tid);
  tile_fetch(po, dout, rb, 0, L, ldq, h * HD, tid);
  if (tid < TILE) {
    pl = tid < L ? lp[tid] : INFINITY;
    pd = tid < L ? dp_[tid] : 0.f;
  }
  const int tr_off = (4 * hf + ((lane & 15) >> 2)) * VP16 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    tile_store_split<X3_F16>(QrH, QrL, pq, tid, KP16);
    tile_store_split<X3_BF16>(QcH, QcL, pq, tid, VP16);
    tile_store_split<X3_BF16>(OrH, OrL, po, tid, KP16);
    tile_store_split<X3_BF16>(OcH, OcL, po, tid, VP16);
    if (tid < TILE) {
      lse_s[tid] = pl;
      dsum_s[tid] = pd;
    }
    __syncthreads();
    if (qt + 1 < nqt) {
      tile_fetch(pq, q, rb, (qt + 1) * TILE, L, ldq, h * HD, tid);
      tile_fetch(po, dout, rb, (qt + 1) * TILE, L, ldq, h * HD, tid);
      if (tid < TILE) {
        const int i = (qt + 1) * TILE + tid;
        pl = i < L ? lp[i] : INFINITY;
        pd = i < L ? dp_[i] : 0.f;
      }
    }
    f32x16 sm = zero16(), sl = zero16(), dpm = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int o16 = li * KP16 + 16 * s + 8 * hf;
      x3_mfma<X3_F16>(*reinterpret_cast<const f16x8 *>(QrH + o16), *reinterpret_cast<const f16x8 *>(QrL + o16), kh[s], kl[s], sm, sl);
      x3_mfma1_bf16(*reinterpret_cast<const bf16x8 *>(OrH + o16), *reinterpret_cast<const bf16x8 *>(OrL + o16), vh[s], vl[s], dpm);
    }
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = acc_row(r, hf);
      p[r] = expf(fmaf(sl[r], X3P<X3_F16>::INV, sm[r]) - lse_s[i]);
      ds[r] = p[r] * (dpm[r] - dsum_s[i]) * scale;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ph, pl8, dh, dl;
      split8<X3_BF16>(p + 8 * s, ph, pl8);
      split8<X3_BF16>(ds + 8 * s, dh, dl);
      const int ob = (16 * s) * VP16 + tr_off;
      x3_mfma1_bf16(ph, pl8, tr_frag_bf16(OcH + ob, VP16), tr_frag_bf16(OcL + ob, VP16), dv0);
      x3_mfma1_bf16(ph, pl8, tr_frag_bf16(OcH + ob + 32, VP16), tr_frag_bf16(OcL + ob + 32, VP16), dv1);
      x3_mfma1_bf16(dh, dl, tr_frag_bf16(QcH + ob, VP16), tr_frag_bf16(QcL + ob, VP16), dk0);
      x3_mfma1_bf16(dh, dl, tr_frag_bf16(QcH + ob + 32, VP16), tr_frag_bf16(QcL + ob + 32, VP16), dk1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = kbase + acc_row(r, hf);
    if (row < L) {
      float *pk_ = dkv + (rb + row) * ldkv + h * HD + li;
      pk_[0] = dk0[r];
      pk_[32] = dk1[r];
      pk_[H * HD] = dv0[r];
      pk_[H * HD + 32] = dv1[r];
    }
  }
}

}  // namespace

bool attention_f32_mfma_ok(int ldq, int ldkv, int ldo, int B, int H) { return (ldkv % 4) == 0 && (ldo % 4) == 0 && ldq > 0 && B <= 65535 && H <= 65535; }

hipError_t launch_attention_f32_mfma(const float *q, int ldq, const float *kv, int ldkv, int B, int L, int H, float *out, int ldo, hipStream_t s,
                                     float *lse_out, bool x3, bool xfmt) {
  if (L < 1 || B < 1 || H < 1) return hipErrorInvalidValue;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
  if (xfmt && !(x3 && (ldq % 4) == 0 && ldo == H * HD && (ldo % 32) == 0)) return hipErrorInvalidValue;
  if (x3 && (ldq % 4) == 0) {   // the split-operand forward (16-byte query loads)
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3((L + 127) / 128, H, B), dim3(256), 0, s, q, ldq, kv, ldkv, L, H, scale_log2e, out, ldo, lse_out, xfmt ? 1 : 0);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel, dim3((L + 127) / 128, H, B), dim3(256), 0, s, q, ldq, kv, ldkv, L, H, scale_log2e, out, ldo, lse_out);
  return hipGetLastError();
}

// q, o, dout, dq: (B, L, H*64);  kv, dkv: (B, L, 2*H*64);  lse, dsum: (B, H, L) scratch
hipError_t launch_attention_bwd(const float *q, const float *kv, const float *o, const float *dout, int B, int L, int H, int D, float *dq, float *dkv,
                                float *lse, float *dsum, hipStream_t s, const float *lse_fwd, bool x3) {
  if (D != HD || L < 1 || B < 1 || H < 1 || H > 65535 || B > 65535) return hipErrorInvalidValue;
  const float scale = 1.0f / sqrtf((float)HD);
  const dim3 grid((L + 127) / 128, H, B);
  if (x3 && lse_fwd) {   // split-operand kernels (they take the forward pass's log-sum-exp)
    hipLaunchKernelGGL(attn_bwd_q_x3_kernel, grid, dim3(256), 0, s, q, kv, o, dout, L, H, scale, dq, dsum, lse_fwd);
    hipLaunchKernelGGL(attn_bwd_kv_x3_kernel, grid, dim3(256), 0, s, q, kv, dout, lse_fwd, dsum, L, H, scale, dkv);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(attn_bwd_q_mfma_kernel, grid, dim3(256), 0, s, q, kv, o, dout, L, H, scale, dq, lse, dsum, lse_fwd);
  hipLaunchKernelGGL(attn_bwd_kv_mfma_kernel, grid, dim3(256), 0, s, q, kv, dout, lse_fwd ? lse_fwd : lse, dsum, L, H, scale, dkv);
  return hipGetLastError();
}

}  // namespace sf
