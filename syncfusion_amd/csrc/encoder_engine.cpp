// Encoder1d engine behind sf_encoder1d_* (include/syncfusion_amd.h).
//
// Restates audio_encoders_pytorch.Encoder1d (reference config
// exp/model/diffusion.yaml:35-43): to_in = ResnetBlock1d(in -> c0, groups 1); per level a strided
// Conv1d(k = 2f+1, stride f, padding f) followed by num_blocks ResnetBlock1d(groups = resnet_groups).
// It runs once per clip (< 1 % of a sampling run) and is bound by streaming the long, thin activations,
// so it stays fp32: thin levels (C < 32) use the direct VALU convolution, wider ones the fp32 MFMA GEMM;
// GroupNorm+SiLU is always the prologue of the following convolution.
#include <exception>
#include <memory>

#include "engine_common.h"

using namespace sf;

namespace {
struct ResBlock {
  int cin = 0, cout = 0, groups = 1;
  float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
  ConvW conv1, conv2, to_out;
  bool has_to_out = false;
};
struct EncLevel {
  int cin = 0, cout = 0, factor = 1;
  ConvW down;
  std::vector<ResBlock> blocks;
};
}  // namespace

struct sf_encoder1d {
  sf_encoder1d_config cfg{};
  DeviceArena arena;
  ResBlock to_in;
  std::vector<EncLevel> levels;
};

namespace {

ResBlock make_res(Packer &pk, const std::string &pre, int cin, int cout, int groups) {
  ResBlock r;
  r.cin = cin;
  r.cout = cout;
  r.groups = groups;
  if (cin % groups || cout % groups) fail(SF_ERR_INVALID, "%s: channels not divisible by groups", pre.c_str());
  r.g1 = pk.copy_f32(pre + ".block1.gn.weight", cin);
  r.b1 = pk.copy_f32(pre + ".block1.gn.bias", cin);
  r.conv1 = pk.conv(pre + ".block1.conv.weight", pk.copy_f32(pre + ".block1.conv.bias", cout), cout, cin, 3, cin % 32 != 0, cin, nullptr);
  r.g2 = pk.copy_f32(pre + ".block2.gn.weight", cout);
  r.b2 = pk.copy_f32(pre + ".block2.gn.bias", cout);
  r.conv2 = pk.conv(pre + ".block2.conv.weight", pk.copy_f32(pre + ".block2.conv.bias", cout), cout, cout, 3, cout % 32 != 0, cout, nullptr);
  r.has_to_out = cin != cout;
  if (r.has_to_out)
    r.to_out = pk.conv(pre + ".to_out.weight", pk.copy_f32(pre + ".to_out.bias", cout), cout, cin, 1, cin % 32 != 0, cin, nullptr);
  for (const ConvW *c : {&r.conv1, &r.conv2, &r.to_out})
    if (c->w && c->direct && c->N > 32) fail(SF_ERR_UNSUPPORTED, "%s: thin convolution with %d outputs", pre.c_str(), c->N);
  return r;
}

struct EncPlan {
  int B = 0, L0 = 0;
  std::vector<int> L;  // length after to_in (index 0) and after each level
  float *buf[3] = {nullptr, nullptr, nullptr};
  float *slab = nullptr;
};

EncPlan make_plan(const sf_encoder1d &e, Workspace &ws, int B, int L0) {
  const auto &c = e.cfg;
  if (B < 1 || L0 < 1) fail(SF_ERR_SHAPE, "bad B/L0");
  EncPlan p;
  p.B = B;
  p.L0 = L0;
  p.L.push_back(L0);
  int64_t maxel = (int64_t)L0 * std::max(c.in_channels, c.channels * c.multipliers[0]);
  int64_t slab = 64;
  int L = L0;
  auto upd = [&](int Lx, int C, int G) {
    GnPlan gp = gn_plan(B, Lx, C);
    slab = std::max<int64_t>(slab, (int64_t)B * gp.nch * G * 2);
  };
  upd(L0, c.in_channels, 1);
  upd(L0, c.channels * c.multipliers[0], 1);
  for (int i = 0; i < c.n_layers; ++i) {
    int f = c.factors[i];
    L = (L - 1) / f + 1;
    p.L.push_back(L);
    int C = c.channels * c.multipliers[i + 1];
    maxel = std::max(maxel, (int64_t)L * C);
    upd(L, C, c.resnet_groups);
  }
  for (int i = 0; i < 3; ++i) p.buf[i] = ws.alloc_n<float>((int64_t)B * maxel);
  p.slab = ws.alloc_n<float>(slab);
  return p;
}

struct EncExec {
  sf_encoder1d &e;
  EncPlan &p;
  hipStream_t s;

  void conv(const ConvW &w, ConvGemmArgs a) {
    a.w = w.w;
    a.bias = w.bias;
    a.N = w.N;
    a.K = w.K;
    a.cin = w.cin;
    a.cin2 = 0;
    a.taps = w.taps;
    a.n_store = w.N;
    if (w.direct) SF_HIP(launch_conv_direct(F32, F32, a, s));
    else SF_HIP(launch_conv_gemm(F32, a, s));
  }
  // out = conv2(silu(gn(conv1(silu(gn(x)))))) + (to_out(x) | x);  x, out, tmp distinct buffers
  void res(const ResBlock &r, const float *x, float *tmp, float *out, int L) {
    const int rows = p.B * L;
    auto gnconv = [&](const ConvW &w, const float *in, int C, const float *g, const float *b, float *o, const float *resid) {
      GnPlan gp = gn_plan(p.B, L, C);
      SF_HIP(launch_gn_stats(F32, in, C, p.B, L, C, r.groups, gp.nch, gp.chunk_rows, p.slab, s));
      ConvGemmArgs a;
      a.src = in;
      a.src_ld = C;
      a.M = rows;
      a.Lout = a.Lsrc = L;
      a.pad = 1;
      a.pro = 1;
      a.G = r.groups;
      a.nch = gp.nch;
      a.chunk_rows = gp.chunk_rows;
      a.stats = p.slab;
      a.gamma = g;
      a.beta = b;
      a.out = o;
      a.out_ld = w.N;
      a.res = resid;
      a.res_ld = w.N;
      conv(w, a);
    };
    const float *resid = x;
    if (r.has_to_out) {
      ConvGemmArgs a;
      a.src = x;
      a.src_ld = r.cin;
      a.M = rows;
      a.Lout = a.Lsrc = L;
      a.out = out;  // park to_out(x) in `out`; conv2 then adds it in place (element-wise, same thread)
      a.out_ld = r.cout;
      conv(r.to_out, a);
      resid = out;
    }
    gnconv(r.conv1, x, r.cin, r.g1, r.b1, tmp, nullptr);
    gnconv(r.conv2, tmp, r.cout, r.g2, r.b2, out, resid);
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

int sf_encoder1d_create(const sf_encoder1d_config *cfg, const sf_tensor *weights, int n_weights, void *stream, sf_encoder1d **out) {
  SF_API_BEGIN
  if (!cfg || !out) fail(SF_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->patch_size != 1) fail(SF_ERR_UNSUPPORTED, "Encoder1d patch_size must be 1");
  if (cfg->n_layers < 1 || cfg->n_layers > SF_MAX_DEPTH) fail(SF_ERR_INVALID, "n_layers out of range");
  std::unique_ptr<sf_encoder1d> e(new sf_encoder1d());
  e->cfg = *cfg;
  WeightMap wm(weights, n_weights);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Packer pk{e->arena, wm, s, F32};
  const int c0 = cfg->channels * cfg->multipliers[0];
  e->to_in = make_res(pk, "to_in", cfg->in_channels, c0, 1);
  int cin = c0;
  for (int i = 0; i < cfg->n_layers; ++i) {
    EncLevel lv;
    lv.cin = cin;
    lv.cout = cfg->channels * cfg->multipliers[i + 1];
    lv.factor = cfg->factors[i];
    const std::string pre = "downsamples." + std::to_string(i);
    const bool direct = cin % 32 != 0;
    if (direct && lv.cout > 32) fail(SF_ERR_UNSUPPORTED, "%s: thin strided convolution with %d outputs", pre.c_str(), lv.cout);
    lv.down = pk.conv(pre + ".down.weight", pk.copy_f32(pre + ".down.bias", lv.cout), lv.cout, cin, 2 * lv.factor + 1, direct, cin, nullptr);
    for (int j = 0; j < cfg->num_blocks[i]; ++j)
      lv.blocks.push_back(make_res(pk, pre + ".blocks." + std::to_string(j), lv.cout, lv.cout, cfg->resnet_groups));
    e->levels.push_back(std::move(lv));
    cin = e->levels.back().cout;
  }
  SF_HIP(hipStreamSynchronize(s));
  *out = e.release();
  return SF_OK;
  SF_API_END
}

void sf_encoder1d_destroy(sf_encoder1d *h) { delete h; }

int64_t sf_encoder1d_workspace_bytes(const sf_encoder1d *h, int B, int L0) {
  try {
    if (!h) fail(SF_ERR_INVALID, "null handle");
    Workspace dry(nullptr, 0);
    make_plan(*h, dry, B, L0);
    return dry.used();
  } catch (const EngineError &) {
    return -1;
  }
}

int sf_encoder1d_forward(sf_encoder1d *h, const float *y, int B, int L0, float *const *xs_out, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!h || !y || !xs_out || !ws) fail(SF_ERR_INVALID, "null argument");
  Workspace w(ws, ws_bytes);
  EncPlan p = make_plan(*h, w, B, L0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  EncExec ex{*h, p, s};
  const auto &c = h->cfg;
  float *a = p.buf[0], *b = p.buf[1], *t = p.buf[2];
  // (B, Cin, L0) channels-first -> channels-last rows
  SF_HIP(launch_cf_to_cl(F32, y, B, c.in_channels, L0, a, c.in_channels, s));
  ex.res(h->to_in, a, t, b, L0);
  std::swap(a, b);
  SF_HIP(launch_cl_to_cf(F32, a, h->to_in.cout, B, h->to_in.cout, L0, xs_out[0], s));
  for (int i = 0; i < c.n_layers; ++i) {
    const EncLevel &lv = h->levels[i];
    const int Lin = p.L[i], Lo = p.L[i + 1];
    ConvGemmArgs g;
    g.src = a;
    g.src_ld = lv.cin;
    g.M = B * Lo;
    g.Lout = Lo;
    g.Lsrc = Lin;
    g.stride = lv.factor;
    g.pad = lv.factor;
    g.out = b;
    g.out_ld = lv.cout;
    ex.conv(lv.down, g);
    std::swap(a, b);
    for (const ResBlock &r : lv.blocks) {
      ex.res(r, a, t, b, Lo);
      std::swap(a, b);
    }
    SF_HIP(launch_cl_to_cf(F32, a, lv.cout, B, lv.cout, Lo, xs_out[1 + i], s));
  }
  return SF_OK;
  SF_API_END
}

}  // extern "C"
