// Version / error / device probes, the onset glue and the op-level test entry points of the C ABI.
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

extern "C" {

const char *sf_version(void) { return "syncfusion_amd 0.1.0 (gfx950)"; }
const char *sf_last_error(void) { return get_error(); }

int sf_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    return 0;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return std::string(prop.gcnArchName).rfind("gfx950", 0) == 0 ? 1 : 0;
}

int sf_onsets_to_track(const float *logits, int N, int T, const int32_t *start_frame, float frame_rate, float sample_rate,
                       float threshold, float *track, int L, void *stream) {
  SF_API_BEGIN
  if (!logits || !track || N < 1 || T < 1 || L < 1 || frame_rate <= 0.f) fail(SF_ERR_INVALID, "bad argument");
  SF_HIP(launch_onsets_to_track(logits, N, T, start_frame, frame_rate, sample_rate, threshold, track, L, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_op_conv1d_cl(int dtype, const void *x, const float *w, const float *bias, const float *gamma, const float *beta, int groups,
                    float eps, const void *residual, int B, int L, int C, int N, int taps, int stride, int pad, int upsample,
                    void *out, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!x || !w || !out || !ws) fail(SF_ERR_INVALID, "null argument");
  if (upsample < 1 || (upsample & (upsample - 1))) fail(SF_ERR_UNSUPPORTED, "upsample must be a power of two");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Workspace wk(ws, ws_bytes);
  const bool direct = (C % 32) != 0;
  if (direct && N > 32) fail(SF_ERR_UNSUPPORTED, "thin convolution with N > 32");
  const int wdt = direct ? F32 : dtype;
  const int K = taps * C;
  void *wp = wk.alloc((int64_t)N * K * dsize(wdt));
  SF_HIP(launch_pack_conv(wdt, w, N, C, 0, C, taps, C, nullptr, wp, K, 0, s));
  ConvGemmArgs a;
  a.src = x;
  a.src_ld = C;
  a.w = wp;
  a.bias = bias;
  a.N = N;
  a.K = K;
  a.cin = C;
  a.taps = taps;
  a.stride = stride;
  a.pad = pad;
  while ((1 << a.up_shift) < upsample) ++a.up_shift;
  a.Lsrc = L;
  a.Lout = (L * upsample + 2 * pad - taps) / stride + 1;
  a.M = B * a.Lout;
  a.out = out;
  a.out_ld = N;
  a.n_store = N;
  a.res = residual;
  a.res_ld = N;
  if (groups > 0) {
    GnPlan gp = gn_plan(B, L, C);
    float *slab = wk.alloc_n<float>((int64_t)B * gp.nch * groups * 2);
    SF_HIP(launch_gn_stats(dtype, x, C, B, L, C, groups, gp.nch, gp.chunk_rows, slab, s));
    a.pro = 1;
    a.G = groups;
    a.nch = gp.nch;
    a.chunk_rows = gp.chunk_rows;
    a.stats = slab;
    a.gamma = gamma;
    a.beta = beta;
    a.eps = eps;
  }
  if (direct) SF_HIP(launch_conv_direct(dtype, dtype, a, s));
  else SF_HIP(launch_conv_gemm(dtype, a, s));
  return SF_OK;
  SF_API_END
}

int sf_op_ln_modulate(int dtype, const void *x, const float *scale_shift, float eps, int B, int L, int C, void *out, void *stream) {
  SF_API_BEGIN
  if (!x || !out) fail(SF_ERR_INVALID, "null argument");
  SF_HIP(launch_ln_modulate(dtype, x, C, scale_shift, 2 * C, eps, B, L, C, out, C, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_op_attention(int dtype, const void *q, const void *kv, int B, int L, int heads, int head_dim, void *out, void *stream) {
  SF_API_BEGIN
  if (!q || !kv || !out) fail(SF_ERR_INVALID, "null argument");
  SF_HIP(launch_attention(dtype, q, heads * head_dim, kv, 2 * heads * head_dim, B, L, heads, head_dim, out, heads * head_dim,
                          static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

}  // extern "C"
