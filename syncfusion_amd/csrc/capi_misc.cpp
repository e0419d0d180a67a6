// Version / error / device probes, the onset glue and the op-level test entry points of the C ABI.
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <cmath>
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

// Measurement aid (bench.py): the shader clock the chip holds WHILE a workload runs.  sf_clock_probe_start puts a one-wave kernel on `stream`
// (use a side stream) that watches the shader-cycle counter against the constant 100 MHz counter for `microseconds`; sf_clock_probe_read
// waits for it and returns MHz.  One probe at a time per process.
static unsigned long long *g_clock_buf = nullptr;
static hipEvent_t g_clock_ev = nullptr;
int sf_clock_probe_start(double microseconds, void *stream) {
  SF_API_BEGIN
  if (microseconds <= 0.0 || microseconds > 5.0e6) fail(SF_ERR_INVALID, "clock probe: 0 < microseconds <= 5e6");
  if (!g_clock_buf) SF_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_clock_buf), 2 * sizeof(unsigned long long), hipHostMallocDefault));
  if (!g_clock_ev) SF_HIP(hipEventCreateWithFlags(&g_clock_ev, hipEventDisableTiming));
  g_clock_buf[0] = g_clock_buf[1] = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  SF_HIP(launch_clock_probe(microseconds, g_clock_buf, s));
  SF_HIP(hipEventRecord(g_clock_ev, s));
  return SF_OK;
  SF_API_END
}
int sf_clock_probe_read(double *mhz_out) {
  SF_API_BEGIN
  if (!mhz_out || !g_clock_ev || !g_clock_buf) fail(SF_ERR_INVALID, "clock probe: nothing started");
  SF_HIP(hipEventSynchronize(g_clock_ev));
  *mhz_out = g_clock_buf[1] ? 100.0 * (double)g_clock_buf[0] / (double)g_clock_buf[1] : 0.0;
  return SF_OK;
  SF_API_END
}

int sf_onsets_to_track(const float *logits, int N, int T, const int32_t *start_frame, float frame_rate, float sample_rate,
                       float threshold, float *track, int L, void *stream) {
  SF_API_BEGIN
  if (!logits || !track || N < 1 || T < 1 || L < 1 || frame_rate <= 0.f) fail(SF_ERR_INVALID, "bad argument");
  SF_HIP(launch_onsets_to_track(logits, N, T, start_frame, frame_rate, sample_rate, threshold, track, L, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_frames_preprocess(const uint8_t *frames, int N, int T, int H, int W, int out_h, int out_w, const float *mean3, const float *std3, float *out,
                         void *stream) {
  SF_API_BEGIN
  if (!frames || !out || !mean3 || !std3 || N < 1 || T < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1) fail(SF_ERR_INVALID, "bad argument");
  for (int i = 0; i < 3; ++i)
    if (!(std3[i] > 0.f)) fail(SF_ERR_INVALID, "std must be positive");
  hipError_t e = launch_frames_preprocess(frames, N, T, H, W, out_h, out_w, mean3, std3, out, static_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) fail(SF_ERR_UNSUPPORTED, "down-scaling factor above 7.5 is not supported");
  SF_HIP(e);
  return SF_OK;
  SF_API_END
}

int sf_times_to_track(const double *times, const int32_t *clip_of, int n_times, double sample_rate, int B, int L, float *track, void *stream) {
  SF_API_BEGIN
  if (!track || B < 1 || L < 1 || n_times < 0 || (n_times > 0 && (!times || !clip_of)) || !(sample_rate > 0)) fail(SF_ERR_INVALID, "bad argument");
  SF_HIP(launch_times_to_track(times, clip_of, n_times, sample_rate, B, L, track, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_cut_prefix_crop(const float *gen, const float *y, int B, int C, int L, int cut_length, float *out, int32_t *first_onset, void *stream) {
  SF_API_BEGIN
  if (!gen || !y || !out || !first_onset || B < 1 || C < 1 || L < 1 || cut_length < 1 || cut_length > L) fail(SF_ERR_INVALID, "bad argument");
  SF_HIP(launch_cut_prefix_crop(gen, y, B, C, L, cut_length, out, first_onset, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

}  // extern "C"

#include <vector>
namespace sf {
void resample_bank(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff, std::vector<float> &bank, int &orig,
                   int &nnew, int &width);
}
struct sf_resampler {
  int orig = 1, nnew = 1, width = 0;
  float *bank = nullptr;
  ~sf_resampler() {
    if (bank) (void)hipFree(bank);
  }
};

extern "C" {

int sf_resampler_create(int orig_freq, int new_freq, int lowpass_filter_width, float rolloff, sf_resampler **out) {
  SF_API_BEGIN
  if (!out || orig_freq < 1 || new_freq < 1 || lowpass_filter_width < 1 || !(rolloff > 0.f && rolloff <= 1.f)) fail(SF_ERR_INVALID, "bad argument");
  *out = nullptr;
  std::vector<float> bank;
  auto *h = new sf_resampler();
  // rolloff arrives as a C float; the reference passes the Python double 0.99 -- use the nearest double of the decimal
  const double ro = std::round((double)rolloff * 1e6) / 1e6;
  resample_bank(orig_freq, new_freq, lowpass_filter_width, ro, bank, h->orig, h->nnew, h->width);
  hipError_t e = hipMalloc(&h->bank, bank.size() * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(h->bank, bank.data(), bank.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    delete h;
    fail(SF_ERR_HIP, "resampler bank upload: %s", hipGetErrorString(e));
  }
  *out = h;
  return SF_OK;
  SF_API_END
}
void sf_resampler_destroy(sf_resampler *h) { delete h; }
int sf_resampler_out_length(const sf_resampler *h, int L) {
  if (!h || L < 0) return -1;
  return (int)(((int64_t)h->nnew * L + h->orig - 1) / h->orig);   // ceil(new * L / orig)
}
int sf_resampler_forward(sf_resampler *h, const float *x, int R, int L, float *out, void *stream) {
  SF_API_BEGIN
  if (!h || !x || !out || R < 1 || L < 1) fail(SF_ERR_INVALID, "bad argument");
  SF_HIP(launch_resample(x, R, L, h->bank, h->orig, h->nnew, h->width, out, sf_resampler_out_length(h, L), static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

// (dgp: the training forward also writes the data-gradient images of the weight, [C][taps][N] fp32 then its split bf16 image, for
// sf_op_conv1d_bwd_cl_p -- one pack launch per weight and step instead of one per GEMM)
// (prepacked: the images come from sf_train_pack_many -- pk_fw / pk_fwx as sf_op_conv1d_train_images says; nothing is packed here.
//  images_out: query only -- which images would the training forward (with / without a data gradient) read?  Nothing is launched.)
enum { IMG_FW = 1, IMG_FWX = 2, IMG_DG = 4, IMG_DGX = 8, IMG_UNPLANNABLE = 16 };
static int conv1d_cl_impl(int dtype, const void *x, const float *w, const float *bias, const float *gamma, const float *beta, int groups,
                          float eps, const void *residual, int B, int L, int C, int N, int taps, int stride, int pad, int upsample,
                          void *out, void *ws, int64_t ws_bytes, void *stream, void *dgp, bool prepacked = false, const float *pk_fw = nullptr,
                          const void *pk_fwx = nullptr, int *images_out = nullptr) {
  SF_API_BEGIN
  if (!images_out && (!x || !w || !out || !ws)) fail(SF_ERR_INVALID, "null argument");
  if (upsample < 1 || (upsample & (upsample - 1))) fail(SF_ERR_UNSUPPORTED, "upsample must be a power of two");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Workspace wk(ws, ws_bytes);
  const bool x3 = dtype == SF_F32X;   // fp32 tensors, products from split fp16 operands (needs N * K * 4 more bytes of workspace)
  if (x3) dtype = F32;
  const bool direct = (C % 32) != 0;
  if (direct && N > 32) fail(SF_ERR_UNSUPPORTED, "thin convolution with N > 32");
  const int wdt = direct ? F32 : dtype;
  const int K = taps * C;
  // ---- the launch, without its weight images ---------------------------------------------------------------------------------------------
  ConvGemmArgs a;
  a.src = x;
  a.src_ld = C;
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
  a.solo = 1;   // op-level entry (the training step's single stream)
  if (groups > 0) {
    GnPlan gp = gn_plan(B, L, C);
    float *slab = images_out ? nullptr : wk.alloc_n<float>((int64_t)B * gp.nch * groups * 2);
    if (!images_out) SF_HIP(launch_gn_stats(dtype, x, C, B, L, C, groups, gp.nch, gp.chunk_rows, slab, s));
    a.pro = 1;
    a.G = groups;
    a.nch = gp.nch;
    a.chunk_rows = gp.chunk_rows;
    a.stats = slab;
    a.gamma = gamma;
    a.beta = beta;
    a.eps = eps;
  }
  // ---- weight images -------------------------------------------------------------------------------------------------------------------------
  const void *wp = w;   // fp32 1x1 convolutions: the PyTorch layout (N, C, 1) IS the GEMM's [N][K] (60 % of the training step's convolutions)
  // (the kernels read weights as 16-byte vectors: a view at a storage offset that is not a multiple of 4 floats is packed like the rest)
  const bool need_fw = !(taps == 1 && wdt == F32 && (reinterpret_cast<uintptr_t>(w) % 16) == 0);
  const bool wx_ok = x3 && !direct && (K % 32) == 0;
  // fragment order for the register-staged kernel, as the engine packs it -- but only when THIS launch would take that kernel (the long rows
  // of the training step go to the macro tiles: 131 of these packs per step were written and never read)
  bool want_rs = false;
  if (wx_ok && groups == 0 && (K % 64) == 0 && K <= 1280 && (N % 32) == 0 && (C % 16) == 0) {
    ConvGemmArgs pa = a;
    pa.w = pa.wx = pa.wfrx = reinterpret_cast<const void *>(16);   // probe
    want_rs = strncmp(conv_gemm_variant_name(dtype, pa), "conv_gemm_rs", 12) == 0;
  }
  void *wx_done = nullptr;
  // which images the training step reads of this weight: the fp32 [N][K] matrix only where the chosen kernel is not a split-operand one
  // (or the fragment-order pack needs it as its source), likewise the fp32 data-gradient matrix
  const bool fw_read = !wx_ok || want_rs || !conv_gemm_reads_split_only(dtype, a);
  const bool dgx_ok = conv1d_dgrad_split_ok(x3, N, taps);
  const bool dg_read = !dgx_ok || !conv_gemm_reads_split_only(F32, conv1d_dgrad_args(nullptr, B, L, C, N, taps, pad, nullptr));
  if (images_out) {
    *images_out = (need_fw && fw_read ? IMG_FW : 0) | (wx_ok ? IMG_FWX : 0) | (dg_read ? IMG_DG : 0) | (dgx_ok ? IMG_DGX : 0) |
                  (want_rs || dtype != F32 || stride != 1 || upsample != 1 || taps > 9 ? IMG_UNPLANNABLE : 0);
    return SF_OK;
  }
  if (prepacked) {   // images written by sf_train_pack_many at the start of the step
    if (want_rs || dtype != F32) fail(SF_ERR_UNSUPPORTED, "prepacked images: not for this launch (sf_op_conv1d_train_images says so)");
    if ((need_fw && fw_read && !pk_fw) || (wx_ok && !pk_fwx)) fail(SF_ERR_INVALID, "prepacked images missing (see sf_op_conv1d_train_images)");
    if (need_fw) wp = fw_read ? pk_fw : nullptr;
    wx_done = const_cast<void *>(pk_fwx);
    if (!wx_ok) wx_done = nullptr;
  } else if (dgp) {   // training forward (fp32 tensors): every image of this weight the step reads, in one launch, and only those
    if (dtype != F32 || stride != 1 || upsample != 1 || taps > 9) fail(SF_ERR_UNSUPPORTED, "data-gradient images: fp32 / fp32x stride-1 convolutions of at most 9 taps");
    float *fw = need_fw && fw_read ? wk.alloc_n<float>((int64_t)N * K) : nullptr;
    if (wx_ok) wx_done = wk.alloc((int64_t)N * K * 4);
    float *dg = static_cast<float *>(dgp);
    void *dgx = dgx_ok ? static_cast<void *>(dg + (int64_t)C * taps * N) : nullptr;
    SF_HIP(launch_pack_train(w, N, C, taps, fw, wx_done, dg_read ? dg : nullptr, dgx, s));
    if (fw) wp = fw;
    else if (need_fw) wp = nullptr;   // nobody reads it
  } else if (need_fw) {
    void *packed = wk.alloc((int64_t)N * K * dsize(wdt));
    if (wx_ok) {   // fp32 matrix and its split image in one pass
      wx_done = wk.alloc((int64_t)N * K * 4);
      SF_HIP(launch_pack_conv_x(w, N, C, taps, static_cast<float *>(packed), wx_done, X3_F16, s));
    } else {
      SF_HIP(launch_pack_conv(wdt, w, N, C, 0, C, taps, C, nullptr, packed, K, 0, s));
    }
    wp = packed;
  }
  if (!direct && dtype != F32 && (K % 64) == 0 && K <= 2048 && (N % 32) == 0 && (C % 16) == 0) {   // as the engine packs it (conv_gemm_rs.hip)
    void *wfr = wk.alloc((int64_t)N * K * dsize(wdt));
    SF_HIP(launch_pack_wfr(dtype, wp, N, K, wfr, s));
    a.wfr = wfr;
  }
  if (wx_done) a.wx = wx_done;
  else if (wx_ok) {
    void *wx = wk.alloc((int64_t)N * K * 4);
    SF_HIP(launch_pack_wx(static_cast<const float *>(wp), N, K, wx, s));
    a.wx = wx;
  }
  a.w = wp;
  if (want_rs) {
    void *wfrx = wk.alloc((int64_t)N * K * 4);
    SF_HIP(launch_pack_wfrx(static_cast<const float *>(wp), N, K, wfrx, s));
    a.wfrx = wfrx;
  }
  if (direct) SF_HIP(launch_conv_direct(dtype, dtype, a, s));
  else SF_HIP(launch_conv_gemm(dtype, a, s));
  return SF_OK;
  SF_API_END
}

int sf_op_conv1d_cl(int dtype, const void *x, const float *w, const float *bias, const float *gamma, const float *beta, int groups,
                    float eps, const void *residual, int B, int L, int C, int N, int taps, int stride, int pad, int upsample,
                    void *out, void *ws, int64_t ws_bytes, void *stream) {
  return conv1d_cl_impl(dtype, x, w, bias, gamma, beta, groups, eps, residual, B, L, C, N, taps, stride, pad, upsample, out, ws, ws_bytes, stream, nullptr);
}

int64_t sf_op_conv1d_dgrad_pack_bytes(int C, int N, int taps) {
  if (C < 1 || N < 1 || taps < 1 || taps > 9) return -1;
  return (int64_t)C * taps * N * 8;   // the fp32 matrix and its split bf16 image
}

int sf_op_conv1d_train_fwd(int dtype, const float *x, const float *w, const float *bias, const float *gamma, const float *beta, int groups, float eps,
                           const float *residual, int B, int L, int C, int N, int taps, int pad, float *out, void *dgrad_pack, int64_t dgrad_pack_bytes,
                           void *ws, int64_t ws_bytes, void *stream) {
  if (dtype != SF_F32 && dtype != SF_F32X) {
    set_error("sf_op_conv1d_train_fwd: dtype must be SF_F32 or SF_F32X");
    return SF_ERR_INVALID;
  }
  if (dgrad_pack && dgrad_pack_bytes < sf_op_conv1d_dgrad_pack_bytes(C, N, taps)) {
    set_error("sf_op_conv1d_train_fwd: dgrad_pack holds %lld bytes, %lld needed", (long long)dgrad_pack_bytes, (long long)sf_op_conv1d_dgrad_pack_bytes(C, N, taps));
    return SF_ERR_INVALID;
  }
  return conv1d_cl_impl(dtype, x, w, bias, gamma, beta, groups, eps, residual, B, L, C, N, taps, 1, pad, 1, out, ws, ws_bytes, stream, dgrad_pack);
}

int sf_op_conv1d_train_images(int dtype, const float *w, int B, int L, int C, int N, int taps, int pad, int groups) {
  if (dtype != SF_F32 && dtype != SF_F32X) {
    set_error("sf_op_conv1d_train_images: dtype must be SF_F32 or SF_F32X");
    return -1;
  }
  int mask = 0;
  const int rc = conv1d_cl_impl(dtype, nullptr, w, nullptr, nullptr, nullptr, groups, 0.f, nullptr, B, L, C, N, taps, 1, pad, 1, nullptr, nullptr, 0, nullptr, nullptr,
                                false, nullptr, nullptr, &mask);
  return rc == SF_OK ? mask : -1;
}

int sf_train_pack_many(const void *desc_dev, int n_items, int total_tiles, void *stream) {
  SF_API_BEGIN
  SF_HIP(launch_pack_train_many(desc_dev, n_items, total_tiles, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_op_conv1d_train_fwd_pk(int dtype, const float *x, const float *w, const float *fw, const void *fwx, const float *bias, const float *gamma,
                              const float *beta, int groups, float eps, const float *residual, int B, int L, int C, int N, int taps, int pad, float *out,
                              void *ws, int64_t ws_bytes, void *stream) {
  if (dtype != SF_F32 && dtype != SF_F32X) {
    set_error("sf_op_conv1d_train_fwd_pk: dtype must be SF_F32 or SF_F32X");
    return SF_ERR_INVALID;
  }
  return conv1d_cl_impl(dtype, x, w, bias, gamma, beta, groups, eps, residual, B, L, C, N, taps, 1, pad, 1, out, ws, ws_bytes, stream, nullptr, true, fw, fwx);
}

int sf_op_gn_silu(int dtype, const void *x, const float *gamma, const float *beta, int groups, float eps, int B, int L, int C, void *out,
                  void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!x || !gamma || !beta || !out) fail(SF_ERR_INVALID, "null argument");
  if (groups < 1 || C % groups) fail(SF_ERR_INVALID, "channels must be divisible by groups");
  SF_HIP(launch_gn_silu_ws(dtype, x, C, B, L, C, groups, gamma, beta, eps, out, C, static_cast<float *>(ws), ws ? ws_bytes / 4 : 0,
                           static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_bench_conv1d(int dtype, int B, int L, int C, int N, int taps, int upsample, int path, int tile, int sk, int iters, float *ms_out) {
  SF_API_BEGIN
  if (!ms_out || iters < 1) fail(SF_ERR_INVALID, "bad argument");
  const bool x3 = dtype == SF_F32X;
  if (x3) dtype = F32;
  const size_t es = dsize(dtype);
  const int K = taps * C, Lout = L * upsample, M = B * Lout;
  void *x = nullptr, *w = nullptr, *out = nullptr, *res = nullptr;
  float *bias = nullptr;
  SF_HIP(hipMalloc(&x, (size_t)B * L * C * es));
  // SF_BENCH_COLD=1: rotate through enough weight copies (> 768 MB) that every launch streams its weights from HBM,
  // as inside a denoising step (430 MB of weights per step do not fit the 256 MB Infinity Cache)
  const size_t wbytes = (size_t)N * K * es;
  int ncopy = 1;
  if (const char *e = tune_env("SF_BENCH_COLD")) {
    if (atoi(e) > 0) ncopy = (int)std::min<size_t>(1024, ((size_t)768 << 20) / wbytes + 1);
  }
  SF_HIP(hipMalloc(&w, wbytes * ncopy));
  SF_HIP(hipMalloc(&out, (size_t)M * N * es));
  SF_HIP(hipMalloc(&res, (size_t)M * N * es));
  SF_HIP(hipMalloc(&bias, N * sizeof(float)));
  // non-trivial bit patterns (zero operands clock higher): bf16/f32 values around +-1
  SF_HIP(hipMemset(x, 0x3c, (size_t)B * L * C * es));
  SF_HIP(hipMemset(w, 0xbc, wbytes * ncopy));
  SF_HIP(hipMemset(res, 0x3d, (size_t)M * N * es));
  SF_HIP(hipMemset(bias, 0, N * sizeof(float)));
  ConvGemmArgs a;
  a.src = x;
  a.src_ld = C;
  a.w = w;
  a.bias = bias;
  a.N = N;
  a.K = K;
  a.cin = C;
  a.taps = taps;
  a.pad = taps / 2;
  while ((1 << a.up_shift) < upsample) ++a.up_shift;
  a.Lsrc = L;
  a.Lout = Lout;
  a.M = M;
  a.out = out;
  a.out_ld = N;
  a.n_store = N;
  a.res = res;
  a.res_ld = N;
  g_conv_gemm_force.path = path;
  g_conv_gemm_force.tile = tile;
  g_conv_gemm_force.sk = sk;
  hipEvent_t e0, e1;
  SF_HIP(hipEventCreate(&e0));
  SF_HIP(hipEventCreate(&e1));
  hipError_t err = hipSuccess;
  unsigned *sink = nullptr;   // the touch kernel's dedicated write sink (never a live buffer)
  void *wfr = nullptr;
  if (dtype != F32 && (K % 64) == 0 && K <= 2048 && (N % 32) == 0 && (C % 16) == 0 && !tune_env("SF_BENCH_NO_WFR")) {
    SF_HIP(hipMalloc(&wfr, wbytes * ncopy));   // fragment-ordered copies, same rotation (the values are random either way)
    SF_HIP(hipMemcpy(wfr, w, wbytes * ncopy, hipMemcpyDeviceToDevice));
    a.wfr = wfr;
  }
  void *wx = nullptr;
  if (x3) {   // split-fp16 images of the same rotation of weight copies
    if (K % 32) fail(SF_ERR_UNSUPPORTED, "fp32x needs K %% 32 == 0");
    SF_HIP(hipMalloc(&wx, wbytes * ncopy));
    for (int c = 0; c < ncopy; ++c)
      SF_HIP(launch_pack_wx(reinterpret_cast<const float *>(static_cast<char *>(w) + (size_t)c * wbytes), N, K, static_cast<char *>(wx) + (size_t)c * wbytes, nullptr));
    a.wx = wx;
  }
  for (int i = 0; i < 3 && err == hipSuccess; ++i) err = launch_conv_gemm(dtype, a, nullptr);
  if (err == hipSuccess) {
    SF_HIP(hipDeviceSynchronize());
    SF_HIP(hipEventRecord(e0, nullptr));
    // SF_BENCH_PREFETCH=n: a `touch` kernel of n workgroups reads this launch's weights right before it (1000 + n: the touch kernel
    // alone) -- what an idle-CU prefetch in the PRECEDING kernel of the chain could buy
    int pre = 0;
    if (const char *e = tune_env("SF_BENCH_PREFETCH")) pre = atoi(e);
    if (pre > 0 && pre % 1000 == 0) fail(SF_ERR_INVALID, "SF_BENCH_PREFETCH=%d: the workgroup count (value mod 1000) must be >= 1", pre);
    if (pre > 0) SF_HIP(hipMalloc(reinterpret_cast<void **>(&sink), 64));
    for (int i = 0; i < iters && err == hipSuccess; ++i) {
      a.w = static_cast<char *>(w) + (size_t)(i % ncopy) * wbytes;
      if (wfr) a.wfr = static_cast<char *>(wfr) + (size_t)(i % ncopy) * wbytes;
      if (wx) a.wx = static_cast<char *>(wx) + (size_t)(i % ncopy) * wbytes;
      if (pre > 0) err = launch_touch(a.w, wbytes, pre % 1000, sink, nullptr);
      if (pre < 1000 && err == hipSuccess) err = launch_conv_gemm(dtype, a, nullptr);
    }
    SF_HIP(hipEventRecord(e1, nullptr));
    SF_HIP(hipEventSynchronize(e1));
    SF_HIP(hipEventElapsedTime(ms_out, e0, e1));
    *ms_out /= (float)iters;
  }
  g_conv_gemm_force = ConvGemmForce();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (void *p : {x, w, out, res, (void *)bias}) (void)hipFree(p);
  if (sink) (void)hipFree(sink);
  if (wfr) (void)hipFree(wfr);
  if (wx) (void)hipFree(wx);
  if (err != hipSuccess) fail(SF_ERR_UNSUPPORTED, "variant not applicable: %s", hipGetErrorString(err));
  return SF_OK;
  SF_API_END
}

int64_t sf_op_resnet_mod_cb_workspace_bytes(int B, int L, int C) {
  Workspace dry(nullptr, 0);
  const int64_t M = (int64_t)B * L;
  dry.alloc(2 * (int64_t)C * C * 3 * 4);   // (sized for SF_F32X: 4 bytes per weight / activation)
  dry.alloc(M * C * 4);
  dry.alloc(M * C * 4);
  dry.alloc((int64_t)(C / 128) * M * C * 4);
  dry.alloc((int64_t)B * 32 * 64 * 2 * 4);
  return dry.used();
}

int sf_op_resnet_mod_cb(int dtype, const void *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *gn1_g,
                        const float *gn1_b, const float *gn2_g, const float *gn2_b, int groups, float eps_gn, const float *scale_shift, float eps_ln,
                        int B, int L, int C, int kb, void *h_out, void *m_out, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (kb != 1 && kb != 2) fail(SF_ERR_INVALID, "kb must be 1 or 2");
  if (!x || !w1 || !b1 || !w2 || !b2 || !gn1_g || !gn1_b || !gn2_g || !gn2_b || !m_out || !ws) fail(SF_ERR_INVALID, "null argument");
  if (!conv_cb_shape_ok(dtype, B, L, C, C, groups) || groups > 64) fail(SF_ERR_UNSUPPORTED, "shape outside the channel-block convolution's coverage");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Workspace wk(ws, ws_bytes);
  const int64_t M = (int64_t)B * L;
  const int S = C / 128 / kb;
  if (kb == 2 && ((C / 128) % 2 || C / groups < 32)) fail(SF_ERR_UNSUPPORTED, "two channel blocks per workgroup need C a multiple of 256 and >= 32 channels per group");
  if (kb == 2 && dtype == SF_F32X) fail(SF_ERR_UNSUPPORTED, "the split-operand form takes one channel block per workgroup");
  const int64_t es = (int64_t)dsize(dtype);
  const int adt = dtype == SF_F32X ? (int)F32 : dtype;   // activations / reducers of the fp32x chain are plain fp32
  char *wp = static_cast<char *>(wk.alloc(2 * (int64_t)C * C * 3 * es));
  void *wp1 = wp, *wp2 = wp + (int64_t)C * C * 3 * es;
  void *act = wk.alloc(M * C * es);
  void *h = h_out ? h_out : wk.alloc(M * C * es);
  float *slab = static_cast<float *>(wk.alloc((int64_t)S * M * C * 4));
  float *stats = static_cast<float *>(wk.alloc((int64_t)B * 32 * 64 * 2 * 4));
  SF_HIP(launch_pack_conv_cb(dtype, w1, C, C, wp1, s));
  SF_HIP(launch_pack_conv_cb(dtype, w2, C, C, wp2, s));
  SF_HIP(launch_gn_silu(adt, x, C, B, L, C, groups, gn1_g, gn1_b, eps_gn, act, C, s));
  ConvCbArgs a;
  a.src = act;
  a.src_ld = C;
  a.wp = wp1;
  a.slab = slab;
  a.B = B;
  a.L = L;
  a.C = a.N = C;
  a.kb = kb;
  SF_HIP(launch_conv_cb(dtype, a, s));
  const CbGnPlan gp = cb_gn_plan(L);
  SF_HIP(launch_cb_reduce_gn(adt, slab, S, B, L, C, b1, h, C, groups, stats, gp, s));
  a.src = h;
  a.wp = wp2;
  a.pro = 1;
  a.G = groups;
  a.nch = gp.nch;
  a.chunk_rows = gp.chunk_rows;
  a.stats = stats;
  a.gamma = gn2_g;
  a.beta = gn2_b;
  a.eps = eps_gn;
  SF_HIP(launch_conv_cb(dtype, a, s));
  SF_HIP(launch_cb_reduce_ln(adt, slab, S, B, L, C, b2, x, C, scale_shift, 2 * C, eps_ln, m_out, C, s));
  return SF_OK;
  SF_API_END
}

int64_t sf_op_inject_prenorm_proj_workspace_bytes(int B, int L, int C, int C2, int N) {
  if (B < 1 || L < 1 || C < 32 || C2 < 0 || N < 8) return -1;
  const int64_t M = (int64_t)B * L, K1 = C + (C2 + 31) / 32 * 32;
  return (int64_t)C * K1 * 2 + (int64_t)N * C * 2 + M * C * 2 + M * (C / 32) * 8 + (int64_t)(2 * N + C) * 4 + 4096;
}

int sf_op_inject_prenorm_proj(int dtype, const void *m, const void *ctx, const float *w_inj, const float *b_inj, const float *gamma,
                              const float *beta, float eps, const float *w_q, int B, int L, int C, int C2, int N, void *z_out, void *q_out,
                              int *fused_out, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!m || !w_inj || !b_inj || !gamma || !beta || !w_q || !z_out || !q_out || !ws) fail(SF_ERR_INVALID, "null argument");
  if (dtype != BF16 && dtype != F16) fail(SF_ERR_UNSUPPORTED, "16-bit dtypes only");
  if (C % 32 || C2 % 32 || N % 32 || (C2 > 0 && !ctx)) fail(SF_ERR_UNSUPPORTED, "C, C2 and N must be multiples of 32");
  const int64_t need = sf_op_inject_prenorm_proj_workspace_bytes(B, L, C, C2, N);
  if (need < 0 || ws_bytes < need) fail(SF_ERR_WORKSPACE, "workspace too small: need %lld bytes", (long long)need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Workspace wk(ws, ws_bytes);
  const int64_t M = (int64_t)B * L;
  const int K1 = C + C2;
  void *w1 = wk.alloc((int64_t)C * K1 * 2), *w2 = wk.alloc((int64_t)N * C * 2), *tmp = wk.alloc(M * C * 2);
  float *rowpart = static_cast<float *>(wk.alloc(M * (C / 32) * 8));
  float *bias2 = static_cast<float *>(wk.alloc((int64_t)N * 4)), *colsum = static_cast<float *>(wk.alloc((int64_t)N * 4));
  SF_HIP(launch_pack_rows(dtype, w_inj, C, K1, K1, nullptr, w1, K1, s));
  SF_HIP(launch_pack_rows(dtype, w_q, N, C, C, gamma, w2, C, s));          // W (g * xhat + b) = (W diag g) xhat + W b
  SF_HIP(launch_fold_bias(w_q, N, C, beta, nullptr, bias2, s));
  SF_HIP(launch_row_sums(dtype, w2, N, C, colsum, s));
  ConvGemmArgs a;   // InjectChannels
  a.src = m;
  a.src_ld = C;
  a.src2 = ctx;
  a.src2_ld = C2;
  a.w = w1;
  a.bias = b_inj;
  a.M = (int)M;
  a.N = a.n_store = C;
  a.K = K1;
  a.cin = C;
  a.cin2 = C2;
  a.taps = 1;
  a.Lout = a.Lsrc = L;
  a.out = z_out;
  a.out_ld = C;
  a.res = m;
  a.res_ld = C;
  ConvGemmArgs q;   // pre-normed projection
  q.src = z_out;
  q.src_ld = C;
  q.w = w2;
  q.bias = bias2;
  q.M = (int)M;
  q.N = q.n_store = N;
  q.K = q.cin = C;
  q.taps = 1;
  q.Lout = q.Lsrc = L;
  q.out = q_out;
  q.out_ld = N;
  ConvGemmArgs ar = a, ql = q;
  ar.rowpart_out = rowpart;
  ar.rowpart_nt = C / 32;
  ql.ln_part = rowpart;
  ql.ln_nt = C / 32;
  ql.ln_eps = eps;
  ql.ln_colsum = colsum;
  ar.mt_ln = ql.mt_ln = 1;   // the op always offers the macro-tile form (the engine never does: kernels.h, ConvGemmArgs::mt_ln)
  const bool fused = conv_gemm_emits_rowpart(dtype, ar) && conv_gemm_ln_ok(dtype, ql);
  if (fused_out) *fused_out = fused ? 1 : 0;
  if (fused) {
    SF_HIP(launch_conv_gemm(dtype, ar, s));
    SF_HIP(launch_conv_gemm_ln(dtype, ql, s));
  } else {
    if (!conv_gemm_supported(dtype, a) || !conv_gemm_supported(dtype, q)) fail(SF_ERR_UNSUPPORTED, "shape outside the GEMM kernels' coverage");
    SF_HIP(launch_conv_gemm(dtype, a, s));
    SF_HIP(launch_ln_modulate(dtype, z_out, C, nullptr, 0, eps, B, L, C, tmp, C, s));
    q.src = tmp;
    SF_HIP(launch_conv_gemm(dtype, q, s));
  }
  return SF_OK;
  SF_API_END
}

// Timing aid: the four launches of the channel-block chain, each averaged over `iters` back-to-back launches on random data
// (ms[0] conv_cb without prologue, ms[1] cb_reduce_gn, ms[2] conv_cb with the GroupNorm+SiLU prologue, ms[3] cb_reduce_ln).
// cold != 0: rotate through enough weight copies that every launch streams its weights from HBM.
int sf_bench_conv_cb(int dtype, int B, int L, int C, int groups, int kb, int cold, int iters, float *ms) {
  SF_API_BEGIN
  if (!ms || iters < 1 || (kb != 1 && kb != 2)) fail(SF_ERR_INVALID, "bad argument");
  if (!conv_cb_shape_ok(dtype, B, L, C, C, groups)) fail(SF_ERR_UNSUPPORTED, "shape outside the channel-block convolution's coverage");
  const int64_t M = (int64_t)B * L;
  const int S = C / 128 / kb;
  const size_t wbytes = (size_t)C * C * 3 * 2;
  int ncopy = 1;
  if (cold) ncopy = (int)std::min<size_t>(256, (768u << 20) / wbytes + 1);
  void *w = nullptr, *x = nullptr, *h = nullptr;
  float *slab = nullptr, *stats = nullptr, *vec = nullptr;
  SF_HIP(hipMalloc(&w, wbytes * ncopy));
  SF_HIP(hipMalloc(&x, M * C * 2));
  SF_HIP(hipMalloc(&h, M * C * 2));
  SF_HIP(hipMalloc(reinterpret_cast<void **>(&slab), (size_t)S * M * C * 4));
  SF_HIP(hipMalloc(reinterpret_cast<void **>(&stats), (size_t)B * 32 * 64 * 2 * 4));
  SF_HIP(hipMalloc(reinterpret_cast<void **>(&vec), (size_t)(4 * C + (size_t)B * 2 * C) * 4));
  SF_HIP(hipMemset(w, 0x11, wbytes * ncopy));     // small finite 16-bit patterns (bf16 / fp16 alike)
  SF_HIP(hipMemset(x, 0x3c, M * C * 2));
  SF_HIP(hipMemset(vec, 0, (size_t)(4 * C + (size_t)B * 2 * C) * 4));
  const CbGnPlan gp = cb_gn_plan(L);
  ConvCbArgs a;
  a.src = x;
  a.src_ld = C;
  a.slab = slab;
  a.B = B;
  a.L = L;
  a.C = a.N = C;
  a.kb = kb;
  a.G = groups;
  a.nch = gp.nch;
  a.chunk_rows = gp.chunk_rows;
  a.stats = stats;
  a.gamma = vec;
  a.beta = vec + C;
  hipEvent_t e0, e1;
  SF_HIP(hipEventCreate(&e0));
  SF_HIP(hipEventCreate(&e1));
  hipError_t err = hipSuccess;
  for (int which = 0; which < 4 && err == hipSuccess; ++which) {
    auto one = [&](int i) -> hipError_t {
      a.wp = static_cast<char *>(w) + (size_t)(i % ncopy) * wbytes;
      switch (which) {
        case 0: a.pro = 0; return launch_conv_cb(dtype, a, nullptr);
        case 1: return launch_cb_reduce_gn(dtype, slab, S, B, L, C, vec + 2 * C, h, C, groups, stats, gp, nullptr);
        case 2: a.pro = 1; return launch_conv_cb(dtype, a, nullptr);
        default: return launch_cb_reduce_ln(dtype, slab, S, B, L, C, vec + 2 * C, x, C, vec + 4 * C, 2 * C, 1e-6f, h, C, nullptr);
      }
    };
    for (int i = 0; i < 3 && err == hipSuccess; ++i) err = one(i);
    if (err != hipSuccess) break;
    SF_HIP(hipDeviceSynchronize());
    SF_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters && err == hipSuccess; ++i) err = one(i);
    SF_HIP(hipEventRecord(e1, nullptr));
    SF_HIP(hipEventSynchronize(e1));
    SF_HIP(hipEventElapsedTime(&ms[which], e0, e1));
    ms[which] /= (float)iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (void *p : {w, x, h, (void *)slab, (void *)stats, (void *)vec}) (void)hipFree(p);
  if (err != hipSuccess) fail(SF_ERR_HIP, "launch failed: %s", hipGetErrorString(err));
  return SF_OK;
  SF_API_END
}

int sf_op_ln_modulate(int dtype, const void *x, const float *scale_shift, float eps, int B, int L, int C, void *out, void *stream) {
  SF_API_BEGIN
  if (!x || !out) fail(SF_ERR_INVALID, "null argument");
  if (scale_shift && (C % 2)) fail(SF_ERR_UNSUPPORTED, "C must be even");
  SF_HIP(launch_ln_modulate(dtype, x, C, scale_shift, 2 * C, eps, B, L, C, out, C, static_cast<hipStream_t>(stream)));
  return SF_OK;
  SF_API_END
}

int sf_op_attention(int dtype, const void *q, const void *kv, int B, int L, int heads, int head_dim, void *out, void *stream) {
  SF_API_BEGIN
  if (!q || !kv || !out) fail(SF_ERR_INVALID, "null argument");
  const bool x3 = dtype == SF_F32X;   // fp32 tensors, products from split fp16 operands
  SF_HIP(launch_attention(x3 ? (int)F32 : dtype, q, heads * head_dim, kv, 2 * heads * head_dim, B, L, heads, head_dim, out, heads * head_dim,
                          static_cast<hipStream_t>(stream), x3));
  return SF_OK;
  SF_API_END
}

}  // extern "C"
