// VideoOnsetNet engine behind sf_onsetnet_* (include/syncfusion_amd.h).
//
// The reference network (main/onset_net.py:12-63, main/resnet.py:36-56,81-114,177-192,234-251) is an
// R(2+1)D-18 whose temporal strides are forced to 1, followed by spatial average pooling and a
// 512-128-1 MLP per frame.  Here every convolution -- (1,7,7) stem, (1,3,3) spatial, (3,1,1) temporal,
// 1x1x1 strided shortcut -- is the same MFMA implicit GEMM over channels-last activations
// (rows = ((n*T + t)*H + h)*W + w), with the eval-mode BatchNorm folded into the weights/bias at build
// time and ReLU / residual add fused in the epilogue.  Odd channel counts (45, 144, 230, 460, 921) are
// zero-padded to multiples of 64 so that K slices never straddle taps.
#include <algorithm>
#include <exception>
#include <memory>

#include "engine_common.h"

using namespace sf;

namespace {

struct Conv3 {
  ConvW w;
  void *wsp = nullptr;   // (1,3,3) stride-1 convolutions of 64 channels: weights in the frame-walk kernel's fragment order (conv_sp.hip)
  void *wtw = nullptr;   // (3,1,1) convolutions with 64 outputs: weights in the temporal-walk kernel's fragment order (conv_tw.hip)
  int cin_real = 0, cin_ld = 0, cout = 0, cout_ld = 0;
  int kt = 1, kh = 1, kw = 1, sh = 1, pt = 0, ph = 0;  // sw == sh, pw == ph, st == 1
};
struct ResBlk {
  Conv3 s1, t1, s2, t2, ds;
  bool has_ds = false;
};

}  // namespace

struct sf_onsetnet {
  int dt = SF_F32;
  DeviceArena arena;
  Conv3 stem_s, stem_t;
  void *stem_wk = nullptr;   // 16-bit types: the stem's weights in the layout of the dedicated kernel (onset_stem.hip)
  std::vector<ResBlk> blocks;  // 8 residual blocks
  ConvW fc0, fc2;
  DebugTaps dbg;
};

namespace {

const int kStagePlanes[4] = {64, 128, 256, 512};
const int kStageStride[4] = {1, 2, 2, 2};
const char *kStageName[4] = {"layer1", "layer2", "layer3", "layer4"};

int midplanes(int in, int planes) { return (in * planes * 27) / (in * 9 + 3 * planes); }  // main/resnet.py:86-87

// conv (+ the BatchNorm that follows it, folded)
// cin_ld / cout_ld: row lengths of the input / output tensors when they are narrower than the default padding (0 = default)
Conv3 make_conv(sf_onsetnet &o, Packer &pk, const std::string &conv_name, const std::string &bn_name, int cin, int cout, int kt,
                int kh, int kw, int sh, int pt, int ph, int cin_ld = 0, int cout_ld = 0) {
  Conv3 c;
  c.cin_real = cin;
  c.cout = cout;
  // multiples of 64 so that every layer but the RGB stem runs on the main (v2) MFMA kernel
  c.cin_ld = cin_ld ? cin_ld : (cin < 32 ? pad_to(cin, 4) : pad_to(cin, 64));
  c.cout_ld = cout_ld ? cout_ld : pad_to(cout, 64);
  c.kt = kt;
  c.kh = kh;
  c.kw = kw;
  c.sh = sh;
  c.pt = pt;
  c.ph = ph;
  float *scale = o.arena.alloc_n<float>(cout), *shift = o.arena.alloc_n<float>(cout);
  SF_HIP(launch_bn_fold(pk.wm.get(bn_name + ".weight", cout), pk.wm.get(bn_name + ".bias", cout), pk.wm.get(bn_name + ".running_mean", cout),
                        pk.wm.get(bn_name + ".running_var", cout), 1e-5f, cout, scale, shift, pk.s));
  c.w = pk.conv(conv_name + ".weight", shift, cout, cin, kt * kh * kw, false, c.cin_ld, scale, 32);
  if (kt == 1 && kh == 3 && kw == 3 && ph == 1 && sh == 1 && c.w.K == 9 * c.cin_ld && conv_sp_ok(o.dt, cin, c.cin_ld, cout, c.cout_ld)) {
    c.wsp = o.arena.alloc((int64_t)conv_sp_weight_elems(cout) * dsize(o.dt));
    SF_HIP(launch_pack_conv_sp(o.dt, c.w.w, cout, c.wsp, pk.s));
  }
  if (kt == 3 && kh == 1 && kw == 1 && pt == 1 && sh == 1 && c.w.K == 3 * c.cin_ld && conv_tw_ok(o.dt, cin, c.cin_ld, cout, c.cout_ld, c.cout_ld)) {
    c.wtw = o.arena.alloc((int64_t)conv_tw_weight_elems(cin) * dsize(o.dt));
    SF_HIP(launch_pack_conv_tw(o.dt, c.w.w, cin, c.cin_ld, c.wtw, pk.s));
  }
  return c;
}

struct OnsetPlan {
  int N = 0, T = 0, H = 0, W = 0;
  void *in = nullptr;          // channels-last input frames
  void *buf[4] = {nullptr, nullptr, nullptr, nullptr};
  float *pooled = nullptr, *hid = nullptr;
};

void out_hw(int Hi, int Wi, const Conv3 &c, int &Ho, int &Wo) {
  Ho = (Hi + 2 * c.ph - c.kh) / c.sh + 1;
  Wo = (Wi + 2 * c.ph - c.kw) / c.sh + 1;
}

OnsetPlan make_plan(const sf_onsetnet &o, Workspace &ws, int N, int T, int H, int W) {
  if (N < 1 || T < 1 || H < 7 || W < 7) fail(SF_ERR_SHAPE, "bad N/T/H/W");
  OnsetPlan p;
  p.N = N;
  p.T = T;
  p.H = H;
  p.W = W;
  const size_t es = dsize(o.dt);
  p.in = ws.alloc((int64_t)N * T * H * W * o.stem_s.cin_ld * es);
  int h, w;
  out_hw(H, W, o.stem_s, h, w);
  int64_t maxel = (int64_t)h * w * std::max(o.stem_s.cout_ld, o.stem_t.cout_ld);
  for (const ResBlk &b : o.blocks) {
    int ho, wo;
    out_hw(h, w, b.s1, ho, wo);
    maxel = std::max(maxel, (int64_t)ho * wo * std::max({b.s1.cout_ld, b.t1.cout_ld, b.s2.cout_ld, b.t2.cout_ld}));
    h = ho;
    w = wo;
  }
  for (int i = 0; i < 4; ++i) p.buf[i] = ws.alloc((int64_t)N * T * maxel * es);
  p.pooled = ws.alloc_n<float>((int64_t)N * T * 512);
  p.hid = ws.alloc_n<float>((int64_t)N * T * 128);
  return p;
}

struct OnsetExec {
  sf_onsetnet &o;
  OnsetPlan &p;
  hipStream_t s;

  void conv(const Conv3 &c, const void *in, int Hi, int Wi, void *out, int &Ho, int &Wo, const void *res, bool relu) {
    out_hw(Hi, Wi, c, Ho, Wo);
    if (c.wsp && !res) {   // layer-1 spatial convolution: frame walk, register-stationary weights, one halo tile per frame
      SF_HIP(launch_conv_sp(o.dt, in, c.cin_ld, c.wsp, c.w.bias, c.cout, out, c.cout_ld, p.N, p.T, Hi, Wi, relu ? 1 : 0, s));
      return;
    }
    if (c.wtw) {   // wide-spatial temporal convolution: frame walk with a three-frame LDS ring (each mid row fetched once, not three times)
      SF_HIP(launch_conv_tw(o.dt, in, c.cin_ld, c.cin_real, c.wtw, c.w.bias, res, c.cout_ld, out, c.cout_ld, p.N, p.T, Ho * Wo, relu ? 1 : 0, s));
      return;
    }
    ConvGemmArgs a;
    a.geom = 1;
    a.src = in;
    a.src_ld = c.cin_ld;
    a.w = c.w.w;
    a.bias = c.w.bias;
    a.N = c.cout;
    a.K = c.w.K;
    a.cin = c.cin_ld;
    a.taps = c.kt * c.kh * c.kw;
    a.M = p.N * p.T * Ho * Wo;
    a.To = a.Ti = p.T;
    a.Ho = Ho;
    a.Wo = Wo;
    a.Hi = Hi;
    a.Wi = Wi;
    a.kt = c.kt;
    a.kh = c.kh;
    a.kw = c.kw;
    a.st = 1;
    a.sh = a.sw = c.sh;
    a.pt = c.pt;
    a.ph = a.pw = c.ph;
    a.out = out;
    a.out_ld = c.cout_ld;
    a.n_store = c.cout_ld;
    a.res = res;
    a.res_ld = c.cout_ld;
    a.act = relu ? 1 : 0;
    a.Lout = a.Lsrc = 1;
    // Column counts that neither 128- nor 192-wide macro tiles cover well (layer 2's 288 mid channels in 320-channel rows: 384 computed
    // columns either way, 25 % of the launch on padding): whole 192-wide tiles first, the remaining <= 128 columns as a second launch
    // (192 + 128 = 320 computed columns).  692 -> ~600 us per convolution at 32 clips.
    static const bool no_split = tune_env("SF_ONSET_NO_NSPLIT") != nullptr;   // A/B aid
    const int q192 = c.cout / 192, rest = c.cout_ld - 192 * q192;
    const int cols_now = std::min((c.cout_ld + 127) / 128 * 128, (c.cout_ld + 191) / 192 * 192);
    if (!no_split && o.dt != F32 && q192 >= 1 && rest > 0 && rest <= 128 && c.cout > 192 * q192 && 192 * q192 + 128 < cols_now) {
      const size_t es = dsize(o.dt);
      ConvGemmArgs a1 = a, a2 = a;
      a1.N = a1.n_store = 192 * q192;
      a2.N = c.cout - 192 * q192;
      a2.n_store = rest;
      a2.w = static_cast<const char *>(a.w) + (size_t)192 * q192 * a.K * es;
      a2.bias = a.bias ? a.bias + 192 * q192 : nullptr;
      a2.out = static_cast<char *>(out) + (size_t)192 * q192 * es;
      a2.res = res ? static_cast<const char *>(res) + (size_t)192 * q192 * es : nullptr;
      SF_HIP(launch_conv_gemm(o.dt, a1, s));
      SF_HIP(launch_conv_gemm(o.dt, a2, s));
      return;
    }
    SF_HIP(launch_conv_gemm(o.dt, a, s));
  }
};

}  // namespace

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

int sf_onsetnet_create(const sf_tensor *weights, int n_weights, int dtype, void *stream, sf_onsetnet **out) {
  SF_API_BEGIN
  if (!out || !weights) fail(SF_ERR_INVALID, "null argument");
  *out = nullptr;
  if (dtype != SF_F32 && dtype != SF_BF16 && dtype != SF_F16) fail(SF_ERR_INVALID, "bad dtype");
  std::unique_ptr<sf_onsetnet> o(new sf_onsetnet());
  o->dt = dtype;
  WeightMap wm(weights, n_weights);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Packer pk{o->arena, wm, s, dtype};
  const std::string m = "net.model.";
  o->stem_s = make_conv(*o, pk, m + "stem.0", m + "stem.1", 3, 45, 1, 7, 7, 2, 0, 3);
  o->stem_t = make_conv(*o, pk, m + "stem.3", m + "stem.4", 45, 64, 3, 1, 1, 1, 1, 0);
  if (dtype != SF_F32 && tune_env("SF_NO_ONSET_STEM") == nullptr) {
    o->stem_wk = o->arena.alloc((int64_t)onset_stem_weight_elems() * dsize(dtype));
    SF_HIP(launch_onset_stem_repack(dtype, o->stem_s.w.w, o->stem_s.cout, o->stem_s.w.K, o->stem_wk, s));
  }
  int cin = 64;
  for (int st = 0; st < 4; ++st) {
    const int planes = kStagePlanes[st];
    for (int bi = 0; bi < 2; ++bi) {
      const int stride = bi == 0 ? kStageStride[st] : 1;
      const int inp = bi == 0 ? cin : planes;
      const std::string pre = m + kStageName[st] + "." + std::to_string(bi);
      ResBlk b;
      const int mid1 = midplanes(inp, planes), mid2 = mid1;  // one midplanes per BasicBlock (main/resnet.py:86-98)
      // Where the temporal convolution runs as the frame walk (conv_tw.hip: 64 outputs, 16-bit types) it is HBM-bound on the mid tensor,
      // and that kernel has no use for rows padded to a multiple of 64: 144 channels travel as 160 instead of 192 (-17 % of the 1.16 GB
      // written by the spatial convolution and read back by the temporal one, per pair, at 32 clips).
      const int mid_ld = conv_tw_ok(dtype, mid1, pad_to(mid1, 32), planes, pad_to(planes, 64), pad_to(planes, 64)) ? pad_to(mid1, 32) : 0;
      b.s1 = make_conv(*o, pk, pre + ".conv1.0.0", pre + ".conv1.0.1", inp, mid1, 1, 3, 3, stride, 0, 1, 0, mid_ld);
      b.t1 = make_conv(*o, pk, pre + ".conv1.0.3", pre + ".conv1.1", mid1, planes, 3, 1, 1, 1, 1, 0, mid_ld, 0);
      b.s2 = make_conv(*o, pk, pre + ".conv2.0.0", pre + ".conv2.0.1", planes, mid2, 1, 3, 3, 1, 0, 1, 0, mid_ld);
      b.t2 = make_conv(*o, pk, pre + ".conv2.0.3", pre + ".conv2.1", mid2, planes, 3, 1, 1, 1, 1, 0, mid_ld, 0);
      b.has_ds = bi == 0 && (stride != 1 || inp != planes);
      if (b.has_ds) b.ds = make_conv(*o, pk, pre + ".downsample.0", pre + ".downsample.1", inp, planes, 1, 1, 1, stride, 0, 0);
      o->blocks.push_back(b);
    }
    cin = planes;
  }
  Packer pf{o->arena, wm, s, F32};  // the per-frame MLP head stays fp32 (rows = N*T, negligible work)
  o->fc0 = pf.linear("fc.0", 128, 512, true);
  o->fc2 = pf.conv("fc.2.weight", pf.copy_f32("fc.2.bias", 1), 1, 128, 1, true, 128, nullptr);
  SF_HIP(hipStreamSynchronize(s));
  *out = o.release();
  return SF_OK;
  SF_API_END
}

void sf_onsetnet_destroy(sf_onsetnet *h) { delete h; }

int64_t sf_onsetnet_workspace_bytes(const sf_onsetnet *h, int N, int T, int H, int W) {
  try {
    if (!h) fail(SF_ERR_INVALID, "null handle");
    Workspace dry(nullptr, 0);
    make_plan(*h, dry, N, T, H, W);
    return dry.used();
  } catch (const EngineError &) {
    return -1;
  }
}

int sf_onsetnet_forward(sf_onsetnet *h, const float *frames, int N, int T, int H, int W, float *logits, void *ws, int64_t ws_bytes,
                        void *stream) {
  SF_API_BEGIN
  if (!h || !frames || !logits || !ws) fail(SF_ERR_INVALID, "null argument");
  Workspace w(ws, ws_bytes);
  OnsetPlan p = make_plan(*h, w, N, T, H, W);
  hipStream_t s = static_cast<hipStream_t>(stream);
  OnsetExec ex{*h, p, s};
  h->dbg.reset();
  SF_HIP(launch_video_to_cl(h->dt, frames, N, 3, T, H, W, p.in, h->stem_s.cin_ld, s));
  void *X = p.buf[0], *M = p.buf[1], *Y = p.buf[2], *R = p.buf[3];
  int hh, ww, h2, w2;
  if (h->stem_wk && h->stem_s.cin_ld == 4 && h->stem_s.cout_ld == 64 && (int64_t)N * T * H * W * 8 < 0x7FFFFFF0ll) {
    out_hw(H, W, h->stem_s, hh, ww);
    SF_HIP(launch_onset_stem(h->dt, p.in, N * T, H, W, h->stem_wk, h->stem_s.w.bias, h->stem_s.cout, M, h->stem_s.cout_ld, s));
  } else {
    ex.conv(h->stem_s, p.in, H, W, M, hh, ww, nullptr, true);
  }
  ex.conv(h->stem_t, M, hh, ww, X, h2, w2, nullptr, true);
  h->dbg.tap("stem", h->dt, X, h->stem_t.cout_ld, (int64_t)N * T * hh * ww, h->stem_t.cout, s);
  int bi = 0;
  for (const ResBlk &b : h->blocks) {
    int ho, wo, t1, t2;
    ex.conv(b.s1, X, hh, ww, M, ho, wo, nullptr, true);
    ex.conv(b.t1, M, ho, wo, Y, t1, t2, nullptr, true);
    ex.conv(b.s2, Y, ho, wo, M, t1, t2, nullptr, true);
    if (b.has_ds) {
      ex.conv(b.ds, X, hh, ww, R, t1, t2, nullptr, false);
      ex.conv(b.t2, M, ho, wo, X, t1, t2, R, true);  // out = relu(conv2 + downsample(x)); X is free once ds ran
    } else {
      ex.conv(b.t2, M, ho, wo, R, t1, t2, X, true);   // out = relu(conv2 + x)
      std::swap(X, R);
    }
    hh = ho;
    ww = wo;
    if (bi % 2 == 1) h->dbg.tap(kStageName[bi / 2], h->dt, X, b.t2.cout_ld, (int64_t)N * T * hh * ww, b.t2.cout, s);
    ++bi;
  }
  // AdaptiveAvgPool3d((None,1,1)) -> (N*T, 512); Linear(512,128)+ReLU; Linear(128,1)
  SF_HIP(launch_spatial_mean(h->dt, X, 512, N * T, hh * ww, 512, p.pooled, s));
  {
    ConvGemmArgs a;
    a.src = p.pooled;
    a.src_ld = 512;
    a.w = h->fc0.w;
    a.bias = h->fc0.bias;
    a.N = 128;
    a.K = 512;
    a.cin = 512;
    a.M = N * T;
    a.out = p.hid;
    a.out_ld = 128;
    a.n_store = 128;
    a.act = 1;
    SF_HIP(launch_conv_gemm(F32, a, s));
  }
  {
    ConvGemmArgs a;
    a.src = p.hid;
    a.src_ld = 128;
    a.w = h->fc2.w;
    a.bias = h->fc2.bias;
    a.N = 1;
    a.K = 128;
    a.cin = 128;
    a.M = N * T;
    a.out = logits;
    a.out_ld = 1;
    a.n_store = 1;
    SF_HIP(launch_conv_direct(F32, F32, a, s));
  }
  return SF_OK;
  SF_API_END
}

int sf_onsetnet_debug_enable(sf_onsetnet *h, float *buf, int64_t cap_floats) {
  if (!h) return SF_ERR_INVALID;
  h->dbg.buf = buf;
  h->dbg.cap = cap_floats;
  h->dbg.reset();
  return SF_OK;
}
int sf_onsetnet_debug_count(const sf_onsetnet *h) { return h ? (int)h->dbg.entries.size() : -1; }
int sf_onsetnet_debug_info(const sf_onsetnet *h, int i, char *name_out, int name_cap, int64_t *offset, int64_t *rows, int32_t *cols) {
  if (!h || i < 0 || i >= (int)h->dbg.entries.size()) return SF_ERR_INVALID;
  const auto &e = h->dbg.entries[i];
  if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", e.name.c_str());
  if (offset) *offset = e.offset;
  if (rows) *rows = e.rows;
  if (cols) *cols = e.cols;
  return SF_OK;
}

}  // extern "C"
