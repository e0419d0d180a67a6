// Training backward, first slice: the op-level C-ABI entry points behind
// syncfusion_amd/autograd.py.  fp32 only -- the reference trains in fp32 (exp/train_diffusion_gh.yaml:87).
#include <algorithm>
#include <exception>

#include "engine_common.h"

using namespace sf;

#define SF_API_BEGIN try {
#define SF_API_END                  \
  }                                 \
  catch (const EngineError &e) {    \
    return e.code;                  \
  }                                 \
  catch (const std::exception &e) { \
    set_error("%s", e.what());      \
    return SF_ERR_INVALID;          \
  }

namespace {

struct BwdPlan {
  float *act = nullptr, *da = nullptr, *wd = nullptr, *wpart = nullptr, *bpart = nullptr, *gpart = nullptr;
  void *wdx = nullptr;   // the dgrad matrix as split fp16 operands (SF_F32X)
  int S = 1, Sb = 1, ldn = 0;
};

BwdPlan plan(Workspace &ws, int B, int L, int C, int N, int taps, int groups) {
  BwdPlan p;
  const int64_t rows = (int64_t)B * L;
  p.ldn = N;   // dgrad reads dy rows of N channels (the MFMA path needs N % 32 == 0, else the direct kernel runs)
  if (groups > 0) p.act = ws.alloc_n<float>(rows * C);
  p.da = ws.alloc_n<float>(rows * C);
  p.wd = ws.alloc_n<float>((int64_t)C * taps * p.ldn);
  p.wdx = ws.alloc((int64_t)C * taps * p.ldn * 4);
  p.S = conv_wgrad_splits(rows, C, N, taps);
  p.wpart = ws.alloc_n<float>((int64_t)p.S * N * taps * C);
  p.Sb = (int)std::min<int64_t>(256, std::max<int64_t>(1, rows * N / 16384));   // >= 16 K elements per slice
  p.bpart = ws.alloc_n<float>((int64_t)p.Sb * N);
  if (groups > 0) p.gpart = ws.alloc_n<float>(gn_silu_bwd_ws_floats(B, L, C, groups));
  return p;
}

}  // namespace

extern "C" {

int64_t sf_op_conv1d_bwd_workspace_bytes(int B, int L, int C, int N, int taps, int groups) {
  try {
    if (B < 1 || L < 1 || C < 1 || N < 1 || taps < 1) fail(SF_ERR_INVALID, "bad shape");
    Workspace dry(nullptr, 0);
    plan(dry, B, L, C, N, taps, groups);
    return dry.used();
  } catch (const EngineError &) {
    return -1;
  }
}

static int conv1d_bwd_impl(int dtype, const float *x, const float *act_saved, const float *stats_saved, const float *w, const float *gamma, const float *beta, int groups, float eps, const float *dy, int B, int L,
                        int C, int N, int taps, int pad, float *dx, float *dw, float *db, float *dgb, void *ws, int64_t ws_bytes, void *stream,
                        const void *dgrad_pack = nullptr /* sf_op_conv1d_train_fwd's images of w: nothing is packed here */,
                        const float *dx_add = nullptr /* GroupNorm convolutions: dx = gradient + dx_add in the GroupNorm backward's own pass */) {
  SF_API_BEGIN
  if (!x || !w || !dy || !ws) fail(SF_ERR_INVALID, "null argument");
  if (dtype != SF_F32 && dtype != SF_F32X) fail(SF_ERR_INVALID, "dtype must be SF_F32 or SF_F32X");
  const bool x3 = dtype == SF_F32X;
  if (!dx && !dw && !db && !dgb) fail(SF_ERR_INVALID, "nothing to compute: dx, dw, db and dgb are all null");
  if (groups > 0 && (!gamma || !beta || !dgb || !dx)) fail(SF_ERR_INVALID, "GroupNorm backward needs gamma, beta, dgb and dx");
  if (dx_add && groups <= 0) fail(SF_ERR_UNSUPPORTED, "dx_add: GroupNorm convolutions only");
  if (taps < 1 || pad < 0 || pad >= taps || 2 * pad != taps - 1) fail(SF_ERR_UNSUPPORTED, "stride-1 'same' convolutions only (2 * pad == taps - 1)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Workspace wk(ws, ws_bytes);
  BwdPlan p = plan(wk, B, L, C, N, taps, groups);
  const float *act = x;
  if (groups > 0 && act_saved) {   // a = SiLU(GroupNorm(x)) kept by the forward pass; its chunk statistics too (else recomputed here)
    if (!stats_saved) SF_HIP(launch_gn_bwd_stats(x, B, L, C, groups, p.gpart, s));
    act = act_saved;
  } else if (groups > 0) {         // recompute a
    SF_HIP(launch_gn_silu_recompute(x, gamma, beta, B, L, C, groups, eps, p.act, p.gpart, s));
    act = p.act;
  }
  // ---- dgrad: da = conv(dy; W flipped and transposed), same padding -------------------------------------------
  // (dx == NULL: the input needs no gradient -- the first convolution on the raw waveform, a frozen trunk -- and without a GroupNorm
  // in front nothing else depends on da: the whole data gradient is skipped; likewise dw == NULL skips the weight gradient)
  if (dx || groups > 0) {
    const bool direct = (N % 32) != 0;
    if (direct && C > 32) fail(SF_ERR_UNSUPPORTED, "dgrad of a thin convolution (N %% 32 != 0) needs C <= 32");
    const bool wx_dg = conv1d_dgrad_split_ok(x3, N, taps);
    ConvGemmArgs a = conv1d_dgrad_args(dy, B, L, C, N, taps, pad, groups > 0 ? p.da : dx);
    if (!wx_dg) a.wx_mode = X3_F16;   // (unused without a split image; the struct's default)
    if (dgrad_pack) {   // images written by sf_op_conv1d_train_fwd: the fp32 matrix only if this launch reads it (the same predicate there)
      p.wd = const_cast<float *>(static_cast<const float *>(dgrad_pack));
      p.wdx = p.wd + (int64_t)C * taps * p.ldn;
      if (wx_dg && conv_gemm_reads_split_only(F32, a)) p.wd = nullptr;
    } else {
      SF_HIP(launch_pack_dgrad(w, N, C, taps, p.ldn, p.wd, s, wx_dg ? p.wdx : nullptr));
    }
    a.w = p.wd;
    if (wx_dg) a.wx = p.wdx;   // products from split bf16 operands
    if (direct) SF_HIP(launch_conv_direct(F32, F32, a, s));
    else SF_HIP(launch_conv_gemm(F32, a, s));
  }
  // ---- wgrad / bias grad ------------------------------------------------------------------------------------------
  // (the bias gradient's slice sums first: their reduction rides on the weight gradient's reducer launch where there is one)
  if (db) SF_HIP(launch_col_sums_part(dy, (int64_t)B * L, N, p.bpart, p.Sb, s));
  bool db_done = false;
  if (dw) SF_HIP(launch_conv_wgrad(dy, act, B, L, C, N, taps, pad, p.wpart, p.S, dw, s, x3 ? X3_BF16 : 0, db ? p.bpart : nullptr, p.Sb, db, &db_done));
  if (db && !db_done) SF_HIP(launch_slices_reduce(p.bpart, p.Sb, N, db, s));
  // ---- GroupNorm + SiLU ----------------------------------------------------------------------------------------------
  if (groups > 0) SF_HIP(launch_gn_silu_bwd(x, p.da, gamma, beta, B, L, C, groups, eps, dx, p.gpart, dgb, s, act_saved ? stats_saved : nullptr, dx_add));
  return SF_OK;
  SF_API_END
}

int sf_op_conv1d_bwd_cl(const float *x, const float *w, const float *gamma, const float *beta, int groups, float eps, const float *dy, int B, int L,
                        int C, int N, int taps, int pad, float *dx, float *dw, float *db, float *dgb, void *ws, int64_t ws_bytes, void *stream) {
  return conv1d_bwd_impl(SF_F32, x, nullptr, nullptr, w, gamma, beta, groups, eps, dy, B, L, C, N, taps, pad, dx, dw, db, dgb, ws, ws_bytes, stream);
}

int sf_op_conv1d_bwd_cl_act(const float *x, const float *act, const float *stats, const float *w, const float *gamma, const float *beta, int groups,
                            float eps, const float *dy, int B, int L, int C, int N, int taps, int pad, float *dx, float *dw, float *db, float *dgb,
                            void *ws, int64_t ws_bytes, void *stream) {
  if (groups > 0 && !act) {
    set_error("sf_op_conv1d_bwd_cl_act: act is null");
    return SF_ERR_INVALID;
  }
  return conv1d_bwd_impl(SF_F32, x, act, stats, w, gamma, beta, groups, eps, dy, B, L, C, N, taps, pad, dx, dw, db, dgb, ws, ws_bytes, stream);
}

int sf_op_conv1d_bwd_cl_x(int dtype, const float *x, const float *act, const float *stats, const float *w, const float *gamma, const float *beta,
                          int groups, float eps, const float *dy, int B, int L, int C, int N, int taps, int pad, float *dx, float *dw, float *db,
                          float *dgb, void *ws, int64_t ws_bytes, void *stream) {
  return conv1d_bwd_impl(dtype, x, act, stats, w, gamma, beta, groups, eps, dy, B, L, C, N, taps, pad, dx, dw, db, dgb, ws, ws_bytes, stream);
}

int sf_op_conv1d_bwd_cl_p(int dtype, const float *x, const float *act, const float *stats, const float *w, const void *dgrad_pack, const float *gamma,
                          const float *beta, int groups, float eps, const float *dy, const float *dx_add, int B, int L, int C, int N, int taps, int pad,
                          float *dx, float *dw, float *db, float *dgb, void *ws, int64_t ws_bytes, void *stream) {
  return conv1d_bwd_impl(dtype, x, act, stats, w, gamma, beta, groups, eps, dy, B, L, C, N, taps, pad, dx, dw, db, dgb, ws, ws_bytes, stream, dgrad_pack,
                         dx_add);
}

int64_t sf_op_gn_silu_train_stats_floats(int B, int L, int C, int groups) {
  if (B < 1 || L < 1 || C < 1 || groups < 1 || C % groups) return -1;
  return gn_bwd_stats_floats(B, L, C, groups);
}

int sf_op_gn_silu_train(const float *x, const float *gamma, const float *beta, int groups, float eps, int B, int L, int C, float *act, float *stats,
                        void *stream) {
  SF_API_BEGIN
  if (!x || !gamma || !beta || !act) fail(SF_ERR_INVALID, "null argument");
  if (groups < 1 || C % groups) fail(SF_ERR_INVALID, "channels must be divisible by groups");
  if (gn_bwd_stats_floats(B, L, C, groups) > 0 && !stats) fail(SF_ERR_INVALID, "stats is null");
  SF_HIP(launch_gn_silu_recompute(x, gamma, beta, B, L, C, groups, eps, act, stats, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int64_t sf_op_length_sums_workspace_bytes(int B, int L, int C) {
  if (B < 1 || L < 1 || C < 1) return -1;
  return (int64_t)B * length_sums_slices(B, L) * C * (int64_t)sizeof(float);
}

int sf_op_length_sums(const float *x, const float *y, int B, int L, int C, float *out, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!x || !out || !ws) fail(SF_ERR_INVALID, "null argument");
  if (B < 1 || L < 1 || C < 1) fail(SF_ERR_INVALID, "B, L and C must be positive");
  const int64_t need = sf_op_length_sums_workspace_bytes(B, L, C);
  if (ws_bytes < need) fail(SF_ERR_WORKSPACE, "workspace too small: need %lld bytes", (long long)need);
  SF_HIP(launch_length_sums(x, y, B, L, C, static_cast<float *>(ws), out, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int64_t sf_op_ln_modulate_bwd_workspace_bytes(int B, int L, int C) {
  if (B < 1 || L < 1 || C < 1) return -1;
  return (int64_t)B * ln_mod_bwd_chunks(L, C) * 2 * C * (int64_t)sizeof(float);
}

static int ln_modulate_bwd_impl(const float *x, const float *scale_shift, const float *dy, const float *dx_add, float eps, int B, int L, int C, float *dx,
                                float *dss, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!x || !dy || !dx || !ws) fail(SF_ERR_INVALID, "null argument");
  if (C < 4 || C > 1024 || (C & (C - 1))) fail(SF_ERR_UNSUPPORTED, "C must be a power of two in [4, 1024] (got %d)", C);
  const int64_t need = sf_op_ln_modulate_bwd_workspace_bytes(B, L, C);
  if (ws_bytes < need) fail(SF_ERR_WORKSPACE, "workspace too small: need %lld bytes", (long long)need);
  SF_HIP(launch_ln_modulate_bwd(x, scale_shift, dy, eps, B, L, C, dx, static_cast<float *>(ws), dss, static_cast<hipStream_t>(stream), dx_add));
  return SF_OK;
  SF_API_END
}

int sf_op_ln_modulate_bwd(const float *x, const float *scale_shift, const float *dy, float eps, int B, int L, int C, float *dx, float *dss, void *ws,
                          int64_t ws_bytes, void *stream) {
  return ln_modulate_bwd_impl(x, scale_shift, dy, nullptr, eps, B, L, C, dx, dss, ws, ws_bytes, stream);
}

int sf_op_ln_modulate_bwd_add(const float *x, const float *scale_shift, const float *dy, const float *dx_add, float eps, int B, int L, int C, float *dx,
                              float *dss, void *ws, int64_t ws_bytes, void *stream) {
  return ln_modulate_bwd_impl(x, scale_shift, dy, dx_add, eps, B, L, C, dx, dss, ws, ws_bytes, stream);
}

int sf_op_attention_bwd(const float *q, const float *kv, const float *out, const float *dout, int B, int L, int heads, int head_dim, float *dq,
                        float *dkv, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!q || !kv || !out || !dout || !dq || !dkv || !ws) fail(SF_ERR_INVALID, "null argument");
  if (head_dim != 64) fail(SF_ERR_UNSUPPORTED, "head_dim must be 64");
  if (B < 1 || L < 1 || heads < 1) fail(SF_ERR_INVALID, "B, L and heads must be positive");
  const int64_t need = (int64_t)2 * B * heads * L * (int64_t)sizeof(float);
  if (ws_bytes < need) fail(SF_ERR_WORKSPACE, "workspace too small: need %lld bytes", (long long)need);
  float *lse = static_cast<float *>(ws), *dsum = lse + (int64_t)B * heads * L;
  SF_HIP(launch_attention_bwd(q, kv, out, dout, B, L, heads, head_dim, dq, dkv, lse, dsum, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

static int attention_fwd_lse_impl(int dtype, const float *q, const float *kv, int B, int L, int heads, int head_dim, float *out, float *lse, void *stream);
int sf_op_attention_fwd_lse(const float *q, const float *kv, int B, int L, int heads, int head_dim, float *out, float *lse, void *stream) {
  return attention_fwd_lse_impl(SF_F32, q, kv, B, L, heads, head_dim, out, lse, stream);
}
int sf_op_attention_fwd_lse_x(int dtype, const float *q, const float *kv, int B, int L, int heads, int head_dim, float *out, float *lse, void *stream) {
  return attention_fwd_lse_impl(dtype, q, kv, B, L, heads, head_dim, out, lse, stream);
}
static int attention_fwd_lse_impl(int dtype, const float *q, const float *kv, int B, int L, int heads, int head_dim, float *out, float *lse, void *stream) {
  SF_API_BEGIN
  if (dtype != SF_F32 && dtype != SF_F32X) fail(SF_ERR_INVALID, "dtype must be SF_F32 or SF_F32X");
  if (!q || !kv || !out || !lse) fail(SF_ERR_INVALID, "null argument");
  if (head_dim != 64) fail(SF_ERR_UNSUPPORTED, "head_dim must be 64");
  if (B < 1 || L < 1 || heads < 1) fail(SF_ERR_INVALID, "B, L and heads must be positive");
  const int hd = heads * head_dim;
  if (!attention_f32_mfma_ok(hd, 2 * hd, hd, B, heads)) fail(SF_ERR_UNSUPPORTED, "shape outside the fp32 matrix-core attention kernel");
  SF_HIP(launch_attention_f32_mfma(q, hd, kv, 2 * hd, B, L, heads, out, hd, static_cast<hipStream_t>(stream), lse, dtype == SF_F32X));
  return SF_OK;
  SF_API_END
}

static int attention_bwd_lse_impl(int dtype, const float *q, const float *kv, const float *out, const float *dout, const float *lse, int B, int L, int heads,
                                  int head_dim, float *dq, float *dkv, void *ws, int64_t ws_bytes, void *stream);
int sf_op_attention_bwd_lse(const float *q, const float *kv, const float *out, const float *dout, const float *lse, int B, int L, int heads, int head_dim,
                            float *dq, float *dkv, void *ws, int64_t ws_bytes, void *stream) {
  return attention_bwd_lse_impl(SF_F32, q, kv, out, dout, lse, B, L, heads, head_dim, dq, dkv, ws, ws_bytes, stream);
}
int sf_op_attention_bwd_lse_x(int dtype, const float *q, const float *kv, const float *out, const float *dout, const float *lse, int B, int L, int heads,
                              int head_dim, float *dq, float *dkv, void *ws, int64_t ws_bytes, void *stream) {
  return attention_bwd_lse_impl(dtype, q, kv, out, dout, lse, B, L, heads, head_dim, dq, dkv, ws, ws_bytes, stream);
}
static int attention_bwd_lse_impl(int dtype, const float *q, const float *kv, const float *out, const float *dout, const float *lse, int B, int L, int heads,
                                  int head_dim, float *dq, float *dkv, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (dtype != SF_F32 && dtype != SF_F32X) fail(SF_ERR_INVALID, "dtype must be SF_F32 or SF_F32X");
  if (!q || !kv || !out || !dout || !lse || !dq || !dkv || !ws) fail(SF_ERR_INVALID, "null argument");
  if (head_dim != 64) fail(SF_ERR_UNSUPPORTED, "head_dim must be 64");
  if (B < 1 || L < 1 || heads < 1) fail(SF_ERR_INVALID, "B, L and heads must be positive");
  const int64_t need = (int64_t)B * heads * L * (int64_t)sizeof(float);
  if (ws_bytes < need) fail(SF_ERR_WORKSPACE, "workspace too small: need %lld bytes", (long long)need);
  float *dsum = static_cast<float *>(ws);
  SF_HIP(launch_attention_bwd(q, kv, out, dout, B, L, heads, head_dim, dq, dkv, nullptr, dsum, static_cast<hipStream_t>(stream), lse, dtype == SF_F32X));
  return SF_OK;
  SF_API_END
}

}  // extern "C"
