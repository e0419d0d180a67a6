// bf16 flash attention on the matrix cores (head dim 64), for the U-Net's self-attention.
//
// One wave owns 32 query rows; a 128-thread workgroup (2 waves) shares 64-key K / V tiles through LDS.
// Swapped products keep everything a softmax needs on the lane that owns the query:
//   S^T = K . Q^T      (32x32x16 MFMA: A = K tile rows from LDS, B = Q^T fragments held in registers)
//        -> lane (q = lane&31, half h) holds 16 of the 32 keys of each S^T tile in its accumulator registers,
//           so the row max / row sum are per-lane loops plus ONE cross-half shuffle;
//   O^T += V^T . P^T   (A = V^T read from a transposed LDS image, B = the P^T accumulator converted to bf16 in
//           place -- the k order of an accumulator-as-operand is permuted (cdna guide section 3), and the V^T
//           fragment is read with the same permutation);
//   O^T has the query on the lane too, so the online-softmax rescale is a per-lane scalar multiply.
// K/V tiles are prefetched into registers while the previous tile is being multiplied (issue-early/write-late).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace sf {
namespace {

constexpr int D = 64, TK = 64, QW = 32;   // head dim, keys per tile, queries per wave
constexpr int WAVES = 2;
constexpr int LDK = D + 8;                // bf16 elements per K-tile row   (144 B: 16-B aligned, conflict-free b128 reads)
constexpr int LDV = TK + 8;               // bf16 elements per V^T-tile row
constexpr int LDO = D + 8;

template <typename T> __global__ __launch_bounds__(128) void attention_mfma_kernel(const T *__restrict__ q, int ldq, const T *__restrict__ kv, int ldkv,
                                                             int L, int H, T *__restrict__ out, int ldo, float scale) {
  __shared__ __attribute__((aligned(16))) T Ks[TK * LDK];
  __shared__ __attribute__((aligned(16))) T Vt[D * LDV];
  __shared__ __attribute__((aligned(16))) T Os[WAVES * QW * LDO];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t rowbase = (size_t)b * L;
  const int q0 = blockIdx.x * (WAVES * QW) + wave * QW;
  const int qi = q0 + fr;
  const bool qvalid = qi < L;

  // Q^T as the B operand of S^T = K.Q^T: lane holds Q[q = fr][d = 16 s + 8 fh + j], pre-scaled by 1/sqrt(D)
  typename Frag16<T>::type qf[4];
  {
    const T *qp = q + (rowbase + (qvalid ? qi : 0)) * ldq + h * D;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Vec16<T> v = ld16<T>(qp + 16 * s + 8 * fh);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (T)((float)v.v[j] * scale);
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int koff = h * D, voff = H * D + h * D;
  // staging: 128 threads x 16 B = 16 rows of 64 bf16 per pass; 4 passes per tile for K and for V
  const int srow = tid >> 3, svec = tid & 7;
  Vec16<T> rk[4], rv[4];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kj = k0 + i * 16 + srow;
      if (kj < L) {
        const T *kp = kv + (rowbase + kj) * ldkv;
        rk[i] = ld16<T>(kp + koff + svec * 8);
        rv[i] = ld16<T>(kp + voff + svec * 8);
      } else {
        rk[i] = zero16<T>();
        rv[i] = zero16<T>();
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = i * 16 + srow;
      st16<T>(Ks + kr * LDK + svec * 8, rk[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) Vt[(svec * 8 + j) * LDV + kr] = rv[i].v[j];   // transposed image: Vt[d][key]
    }
  };

  prefetch(0);
  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();   // previous tile consumed by every wave
    stage();
    __syncthreads();
    if (k0 + TK < L) prefetch(k0 + TK);

    // ---- S^T tiles: keys 32 t .. 32 t + 31 on the accumulator rows, queries on the lanes ----
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        typename Frag16<T>::type kf = *reinterpret_cast<const typename Frag16<T>::type *>(Ks + (32 * t + fr) * LDK + 16 * s + 8 * fh);
        st[t] = mfma32x16(kf, qf[s], st[t]);
      }
    }
    // ---- online softmax (fp32) ----
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (key >= L) st[t][r] = -INFINITY;
        tmax = fmaxf(tmax, st[t][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __expf(mrun - mnew);
    float psum = 0.f;
    typename Frag16<T>::type pf[4];   // P^T as B operand: k-step (t, s) takes accumulator registers 8 s .. 8 s + 7 of tile t
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __expf(st[t][r] - mnew);
        psum += pv;
        pf[2 * t + (r >> 3)][r & 7] = (T)pv;
      }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    // ---- O^T += V^T . P^T : A element j of lane half fh is key 32 t + 16 s + 8 (j>>2) + 4 fh + (j&3) ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const T *vp = Vt + (32 * i + fr) * LDV + 16 * ks + 4 * fh;
        typedef __attribute__((ext_vector_type(4))) T x4_t;
        x4_t lo = *reinterpret_cast<const x4_t *>(vp);
        x4_t hi = *reinterpret_cast<const x4_t *>(vp + 8);
        typename Frag16<T>::type vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = lo[j];
          vf[4 + j] = hi[j];
        }
        o[i] = mfma32x16(vf, pf[ks], o[i]);
      }
  }
  lrun += __shfl_xor(lrun, 32, 64);
  const float inv = 1.0f / lrun;
  // ---- O^T (d on rows, q on lanes) -> LDS [q][d] -> 16-byte row stores ----
  T *os = Os + wave * QW * LDO;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) os[fr * LDO + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh] = (T)(o[i][r] * inv);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes have landed before it reads them back
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
    const int qq = q0 + row;
    if (qq < L) st16<T>(out + (rowbase + qq) * ldo + h * D + c8 * 8, ld16<T>(os + row * LDO + c8 * 8));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Long sequences (the reference's 2^18-sample clips: L = 2048 / 1024 at depths 4 / 5): same products, but
//   * 4 waves = 128 queries share every K / V tile (half the L2 -> LDS traffic per query of the 2-wave kernel);
//   * V is staged ROW-major with 16-byte stores and its transposed fragments are gathered by the hardware
//     (`ds_read_b64_tr_b16`, gfx950: a 16-lane group reads 4 keys x 16 head dims and every lane receives one head dim of the
//     4 keys) -- the 2-wave kernel writes the transposed image with 32 two-byte stores per thread and tile.  The accumulator-
//     as-operand k order (element j of half h = key 16 s + 8 (j >> 2) + 4 h + (j & 3)) is two blocks of 4 consecutive keys:
//     exactly two transposed reads per fragment.  Row pitch 192 B: the 4 rows of a block fall in 4 disjoint 16-bank ranges;
//   * scores are kept in log2 units (q pre-scaled by log2(e) / 8): the exponential is one v_exp_f32.
// ---------------------------------------------------------------------------------------------------------------
constexpr int W4 = 4;
constexpr int LDV4 = 96;   // bf16 elements per row of the row-major V tile (192 B)
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

template <typename T> __global__ __launch_bounds__(256) void attention_mfma4_kernel(const T *__restrict__ q, int ldq, const T *__restrict__ kv, int ldkv,
                                                              int L, int H, T *__restrict__ out, int ldo, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) T Ks[TK * LDK];
  __shared__ __attribute__((aligned(16))) T Vs[TK * LDV4];
  __shared__ __attribute__((aligned(16))) T Os[W4 * QW * LDO];
  using frag = typename Frag16<T>::type;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t rowbase = (size_t)b * L;
  const int q0 = blockIdx.x * (W4 * QW) + wave * QW;
  const int qi = q0 + fr;
  const bool qvalid = qi < L;

  frag qf[4];
  {
    const T *qp = q + (rowbase + (qvalid ? qi : 0)) * ldq + h * D;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Vec16<T> v = ld16<T>(qp + 16 * s + 8 * fh);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (T)((float)v.v[j] * scale_log2e);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int koff = h * D, voff = H * D + h * D;
  const int srow = tid >> 3, svec = tid & 7;   // 256 threads x 16 B = 32 rows of 64 elements per pass
  Vec16<T> rk[2], rv[2];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kj = k0 + i * 32 + srow;
      if (kj < L) {
        const T *kp = kv + (rowbase + kj) * ldkv;
        rk[i] = ld16<T>(kp + koff + svec * 8);
        rv[i] = ld16<T>(kp + voff + svec * 8);
      } else {
        rk[i] = zero16<T>();
        rv[i] = zero16<T>();
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = i * 32 + srow;
      st16<T>(Ks + kr * LDK + svec * 8, rk[i]);
      st16<T>(Vs + kr * LDV4 + svec * 8, rv[i]);
    }
  };
  // transposed-read lane geometry: group of 16 lanes -> (head-dim block 16 * ((lane >> 4) & 1), keys +4 for the upper half-wave);
  // lane 4 q + p of the group addresses key row q, head dims 4 p .. 4 p + 3 of the block
  const int tr_off = (4 * fh + ((lane & 15) >> 2)) * LDV4 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  prefetch(0);
  for (int k0 = 0; k0 < L; k0 += TK) {
    __syncthreads();
    stage();
    __syncthreads();
    if (k0 + TK < L) prefetch(k0 + TK);

    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const frag kf = *reinterpret_cast<const frag *>(Ks + (32 * t + fr) * LDK + 16 * s + 8 * fh);
        st[t] = mfma32x16(kf, qf[s], st[t]);
      }
    }
    float tmax = -INFINITY;
    if (k0 + TK > L) {   // ragged last tile only
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * fh >= L) st[t][r] = -INFINITY;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[t][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    float psum = 0.f;
    frag pf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(st[t][r] - mnew);
        psum += pv;
        pf[2 * t + (r >> 3)][r & 7] = (T)pv;
      }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const T *vp = Vs + tr_off + (16 * ks) * LDV4 + 32 * i;
        const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)(vp));
        const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)(vp + 8 * LDV4));
        typedef short v8i16 __attribute__((ext_vector_type(8)));
        const v8i16 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[i] = mfma32x16(__builtin_bit_cast(frag, both), pf[ks], o[i]);
      }
  }
  lrun += __shfl_xor(lrun, 32, 64);
  const float inv = 1.0f / lrun;
  T *os = Os + wave * QW * LDO;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) os[fr * LDO + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh] = (T)(o[i][r] * inv);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
    const int qq = q0 + row;
    if (qq < L) st16<T>(out + (rowbase + qq) * ldo + h * D + c8 * 8, ld16<T>(os + row * LDO + c8 * 8));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Key-split variant for SHORT sequences (the U-Net's L = 44 ... 352 at batch 4-8: a few hundred 2-wave workgroups of
// the kernel above leave most SIMDs empty and every latency exposed).  A 256-thread workgroup owns ONE 32-query tile;
// its four waves take the 32-key tiles round-robin, each with wave-private LDS staging (no workgroup barrier in the
// loop), and the four partial (max, sum, O) states are merged once through LDS (flash-decoding).  Same products,
// layouts and operand permutation as above with TK = 32.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KS = 4, TK2 = 32;
constexpr int LDK2 = D + 8;
constexpr int LDVR = 96;                         // row pitch of the ROW-major V tile (192 B, as in the 4-wave kernel below)
constexpr int WSTG = TK2 * LDK2 + TK2 * LDVR;   // bf16 elements of one wave's staging area (K tile | V tile)
typedef short v4i16k __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16k lds_v4i16k;

template <typename T> __global__ __launch_bounds__(256) void attention_ksplit_kernel(const T *__restrict__ q, int ldq, const T *__restrict__ kv, int ldkv,
                                                               int L, int H, T *__restrict__ out, int ldo, float scale) {
  // staging: KS x WSTG bf16 (43 KB); merge (aliases it after a barrier): KS x 32 x (D + 1) floats + KS x 32 x 2
  __shared__ __attribute__((aligned(16))) unsigned char smem[KS * WSTG * 2 > KS * 32 * (D + 4) * 4 ? KS * WSTG * 2 : KS * 32 * (D + 4) * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const size_t rowbase = (size_t)b * L;
  const int q0 = blockIdx.x * 32;
  const int qi = q0 + fr;
  const bool qvalid = qi < L;
  T *Ks = reinterpret_cast<T *>(smem) + (size_t)wave * WSTG;
  T *Vs = Ks + TK2 * LDK2;

  typename Frag16<T>::type qf[4];
  {
    const T *qp = q + (rowbase + (qvalid ? qi : 0)) * ldq + h * D;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Vec16<T> v = ld16<T>(qp + 16 * s + 8 * fh);
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[s][j] = (T)((float)v.v[j] * scale);
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int koff = h * D, voff = H * D + h * D;
  const int srow = lane >> 3, svec = lane & 7;   // one wave stages 8 rows x 64 d per pass, 4 passes per 32-key tile
  const int ntk = (L + TK2 - 1) / TK2;
  Vec16<T> rk[4], rv[4];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kj = k0 + i * 8 + srow;
      if (kj < L) {
        const T *kp = kv + (rowbase + kj) * ldkv;
        rk[i] = ld16<T>(kp + koff + svec * 8);
        rv[i] = ld16<T>(kp + voff + svec * 8);
      } else {
        rk[i] = zero16<T>();
        rv[i] = zero16<T>();
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = i * 8 + srow;
      st16<T>(Ks + kr * LDK2 + svec * 8, rk[i]);
      st16<T>(Vs + kr * LDVR + svec * 8, rv[i]);   // row-major: the transposed fragments are gathered by ds_read_b64_tr_b16 (until round 3
    }                                              // 32 two-byte stores per thread wrote V^T: 65 % of this kernel's LDS cycles were bank conflicts)
  };
  // transposed-read lane geometry (see attention_mfma4_kernel): lane 4 q + p of a 16-lane group addresses key row q, head dims 4 p ... 4 p + 3
  const int tr_off = (4 * fh + ((lane & 15) >> 2)) * LDVR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  if (wave < ntk) prefetch(wave * TK2);
  for (int t = wave; t < ntk; t += KS) {
    const int k0 = t * TK2;
    __builtin_amdgcn_s_waitcnt(0xC07F);   // this wave's LDS reads of the previous tile are done (lgkmcnt 0)
    __builtin_amdgcn_wave_barrier();
    stage();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    if (t + KS < ntk) prefetch((t + KS) * TK2);

    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      typename Frag16<T>::type kf = *reinterpret_cast<const typename Frag16<T>::type *>(Ks + fr * LDK2 + 16 * s + 8 * fh);
      st = mfma32x16(kf, qf[s], st);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (key >= L) st[r] = -INFINITY;
      tmax = fmaxf(tmax, st[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);   // finite: a tile always holds at least one valid key
    const float alpha = __expf(mrun - mnew);
    float psum = 0.f;
    typename Frag16<T>::type pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __expf(st[r] - mnew);
      psum += pv;
      pf[r >> 3][r & 7] = (T)pv;
    }
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const T *vp = Vs + tr_off + (16 * ks) * LDVR + 32 * i;
        const v4i16k lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16k *)(vp));
        const v4i16k hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16k *)(vp + 8 * LDVR));
        typedef short v8i16k __attribute__((ext_vector_type(8)));
        const v8i16k both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[i] = mfma32x16(__builtin_bit_cast(typename Frag16<T>::type, both), pf[ks], o[i]);
      }
  }
  lrun += __shfl_xor(lrun, 32, 64);

  // ---- merge the four key ranges: (m, l, O^T) per wave -> LDS [wave][q][d] fp32 ---------------------------------------
  __syncthreads();   // every wave is done with its staging area (the merge buffers alias it)
  float *mo = reinterpret_cast<float *>(smem);            // [KS][32][D + 1]
  float *ml = mo + KS * 32 * (D + 1);                     // [KS][32][2] = (m, l)
  {
    float *om = mo + ((size_t)wave * 32 + fr) * (D + 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) om[32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh] = o[i][r];
    if (fh == 0) {
      ml[(wave * 32 + fr) * 2] = mrun;
      ml[(wave * 32 + fr) * 2 + 1] = lrun;
    }
  }
  __syncthreads();
  {
    const int qq = tid >> 3, c8 = tid & 7;   // 32 queries x 8 chunks of 8 d
    float mw[KS], lw[KS], mstar = -INFINITY;
#pragma unroll
    for (int w = 0; w < KS; ++w) {
      mw[w] = ml[(w * 32 + qq) * 2];
      lw[w] = ml[(w * 32 + qq) * 2 + 1];
      mstar = fmaxf(mstar, mw[w]);
    }
    float lsum = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int w = 0; w < KS; ++w) {
      const float sc = lw[w] > 0.f ? __expf(mw[w] - mstar) : 0.f;   // a wave without tiles contributes nothing
      lsum += lw[w] * sc;
      const float *ow = mo + ((size_t)w * 32 + qq) * (D + 1) + c8 * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(ow[j], sc, acc[j]);
    }
    const float inv = 1.0f / lsum;
    if (q0 + qq < L) {
      Vec16<T> ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) ov.v[j] = (T)(acc[j] * inv);
      st16<T>(out + (rowbase + q0 + qq) * ldo + h * D + c8 * 8, ov);
    }
  }
}

}  // namespace

template <typename T>
static hipError_t attention_mfma_go(const void *q, int ldq, const void *kv, int ldkv, int B, int L, int H, void *out, int ldo, hipStream_t s) {
  // short sequences: the 2-wave kernel would launch fewer waves than the chip has SIMDs -> split the keys across waves
  const long waves_plain = (long)((L + WAVES * QW - 1) / (WAVES * QW)) * H * B * WAVES;
  const long wgs4 = (long)((L + W4 * QW - 1) / (W4 * QW)) * H * B;   // workgroups of the 4-wave kernel
  static const long wgs4_min = [] {   // tuning hook: from this many 4-wave workgroups on the long-sequence kernel runs instead of the key split
    const char *e = tune_env("SF_ATTN_WGS4_MIN");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 128L;
  }();
  const bool long_kernel = L >= 256 && wgs4 >= wgs4_min;
  static const long ksplit_waves = [] {   // tuning hook: the key-split kernel runs while the 2-wave kernel would launch fewer waves than this
    const char *e = tune_env("SF_ATTN_KSPLIT_WAVES");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 1024L;   // (2048 until round 5: at 32 evaluations per branch the 176-position level keeps 6 key-split workgroups per head
                                // re-reading K and V; configs[2] +1.0 %, batch 32 without guidance +-0, profiles/r5_f_ab_attn_ksplit.txt)
  }();
  if (!long_kernel && waves_plain < ksplit_waves && L <= 4096) {
    dim3 g2((L + 31) / 32, H, B);
    hipLaunchKernelGGL((attention_ksplit_kernel<T>), g2, dim3(256), 0, s, static_cast<const T *>(q), ldq, static_cast<const T *>(kv), ldkv,
                       L, H, static_cast<T *>(out), ldo, 1.0f / sqrtf((float)D));
    return hipGetLastError();
  }
  static const bool two_wave = tune_env("SF_ATTN_2WAVE") != nullptr;   // tuning hook: the 2-wave kernel for every length
  if (!two_wave && L >= 256) {
    dim3 g4((L + W4 * QW - 1) / (W4 * QW), H, B);
    hipLaunchKernelGGL((attention_mfma4_kernel<T>), g4, dim3(256), 0, s, static_cast<const T *>(q), ldq, static_cast<const T *>(kv), ldkv, L, H,
                       static_cast<T *>(out), ldo, 1.4426950408889634f / sqrtf((float)D));
    return hipGetLastError();
  }
  dim3 grid((L + WAVES * QW - 1) / (WAVES * QW), H, B);
  hipLaunchKernelGGL((attention_mfma_kernel<T>), grid, dim3(128), 0, s, static_cast<const T *>(q), ldq, static_cast<const T *>(kv), ldkv,
                     L, H, static_cast<T *>(out), ldo, 1.0f / sqrtf((float)D));
  return hipGetLastError();
}

hipError_t launch_attention_mfma(int dt, const void *q, int ldq, const void *kv, int ldkv, int B, int L, int H, int Dh, void *out, int ldo,
                                 hipStream_t s) {
  if (Dh != D || L <= 0 || dt == F32) return hipErrorInvalidValue;
  return dt == F16 ? attention_mfma_go<f16>(q, ldq, kv, ldkv, B, L, H, out, ldo, s) : attention_mfma_go<bf16>(q, ldq, kv, ldkv, B, L, H, out, ldo, s);
}

}  // namespace sf
