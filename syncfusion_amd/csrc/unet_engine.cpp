// U-Net denoiser + v-sampler engine behind sf_unet_* / sf_vsample (include/syncfusion_amd.h).
//
// Restates audio_diffusion_pytorch.UNetV0 (a-unet XUNet; SURVEY.md appendix A.3, config
// exp/model/diffusion.yaml:11-33 of the reference) as a static launch plan over channels-last
// activations:
//   * every Conv1d (k=3 ResnetItem convs, patchify down-convs, nearest-upsample+conv3 up-convs, the 1x1
//     InjectChannels conv over cat[x, ctx], attention projections, per-clip Linear layers) is ONE kernel
//     family: the MFMA implicit GEMM (conv_gemm.hip) -- or the VALU direct conv for thin layers (C < 32);
//   * GroupNorm+SiLU is the A-operand prologue of the following conv (statistics from gn_stats);
//   * Modulation = ln_modulate; all 34+8 Modulation / SkipModulate Linear layers of a step are one GEMM;
//   * the cross-attention over the single CLAP token collapses to a per-clip bias (softmax over one key is
//     exactly 1), computed once per sample() call and added in the preceding conv's epilogue;
//   * LayerNorm affines of the attention pre-norms are folded into the q/kv projection weights;
//   * with classifier-free guidance the conditional and unconditional evaluations run as one 2B batch.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>

#include "engine_common.h"

using namespace sf;

namespace {

struct Group {
  float *gn1_g = nullptr, *gn1_b = nullptr, *gn2_g = nullptr, *gn2_b = nullptr;
  ConvW conv1, conv2, inject, qkv, attn_out, cross_out;
  float *qkv_colsum = nullptr;   // sum_k of every packed qkv row: LayerNorm of the raw input applied to the accumulator
  int mod_off = 0;  // column of [scale | shift] inside mod_all
  bool attn = false, cross = false;
  int ca_off = 0;   // column of the collapsed cross-attention bias inside ca_all
  int ca_idx = 0;   // index of this group's v projection inside wv_cat
};

struct Block {
  int C = 0, cin = 0, factor = 1, ctx = 0, ctx_ld = 0, up_shift = 0;
  ConvW down, up;
  bool up_transposed = false;   // ConvTranspose1d(kernel = stride = factor): a plain GEMM onto the (rows, factor * cin) view of the output
  int skip_off = 0, skip_cols = 0;   // SkipModulate scale inside mod_all (tiled `factor` times for the transposed up path)
  std::vector<Group> down_items, up_items;
};

struct Level {
  int L = 0, C = 0;
  int64_t rows = 0;
  void *buf[3] = {nullptr, nullptr, nullptr};
  void *qkv = nullptr, *ao = nullptr, *ctx = nullptr, *act = nullptr;
  bool cb = false;   // the item heads of this level run as the channel-block split-K chain (conv_cb.hip)
  int cb_kb = 1;     // ... with this many 128-channel blocks per workgroup (2 halves the partial slabs)
};

struct Plan {  // everything carved out of the caller's workspace for one (B, L0, two_pass)
  int B = 0, Bt = 0, L0 = 0;
  bool two = false;
  std::vector<Level> lv;
  float *xs = nullptr;         // sampler state (B, L0, in_channels) the step graph works on
  float *x2 = nullptr, *vout = nullptr, *mod_all = nullptr, *ca_all = nullptr, *slab = nullptr, *emb2 = nullptr;
  float *sched = nullptr;  // [steps][4] = (alpha_i, beta_i, alpha_{i+1}, beta_{i+1})
  float *sigs = nullptr;   // [steps] schedule sigmas, then [Bt] per-row sigmas of a single forward
  void *four = nullptr, *f1 = nullptr, *f2 = nullptr, *sf = nullptr, *emb_t = nullptr, *xhat_e = nullptr, *v_all = nullptr;
  int *step = nullptr;
  int nbr = 1;                 // clip-parallel branches
  int nbr_total = 1;           // ... of the whole plan (a branch view keeps it: the solo hint of its launches)
  int64_t slab_stride = 0;     // floats of GroupNorm scratch per branch
  int64_t slab_half = 0;       // second statistics slab of a branch starts here
  float *rowpart = nullptr;    // per-row LayerNorm partials of the wide levels: two buffers per branch (y and z of an item)
  int64_t rowpart_half = 0, rowpart_stride = 0;
  float *cbslab = nullptr;     // fp32 partial slabs of the channel-block convolutions: one buffer per branch
  int64_t cbslab_stride = 0;
  float *gnpart = nullptr;     // GroupNorm tile sums a producing GEMM leaves for the next item's first convolution: one buffer per branch
  int64_t gnpart_stride = 0;
  // modulation vectors: per clip ([Bt][mod_ld], mod_stride = mod_ld) for a single forward, or ONE row shared by all
  // clips (mod_stride = 0) inside the sampler, where sigma is the same for every clip and the rows of all steps are
  // computed once per call (mod_steps [steps][mod_ld])
  int mod_stride = 0;
  float *mod_steps = nullptr;
  int steps_cap = 0;
};

}  // namespace

struct sf_unet {
  sf_unet_config cfg{};
  int dt = SF_F32;
  bool x3 = false;   // SF_F32X: fp32 activations and kernels, every GEMM on split fp16 operands (ConvW::wx)
  DeviceArena arena;
  float *fourier_w = nullptr;
  int half = 0, four_ld = 0, mf = 0, hd = 0;
  ConvW lin0, mlp0, mlp1, mod, wv_cat;
  int mod_cols = 0, mod_ld = 0, ca_cols = 0, ca_ld = 0, n_ca = 0;
  float *fixed_emb = nullptr;
  CrossOutItem *ca_items = nullptr;   // every cross-attention output projection, for the one-launch collapse (misc.hip)
  int2 *ca_blocks = nullptr;
  int ca_nblocks = 0;
  std::vector<Block> blocks;
  DebugTaps dbg;
  int launches = 0;
  int graph_captures = 0;   // step graphs captured + instantiated so far (diagnostic: a cached shape must not add to it)
  bool listing = false;
  // per-launch HIP-event timing (bench.py's roofline leg): events are recorded on the launch stream
  struct ProfRec {
    const char *label;
    double flops, bytes;
    hipEvent_t e0, e1;
    float ms;
    int depth;   // U-Net depth the launch belongs to (-1: per-step features / sampler glue)
  };
  bool prof_on = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  hipEvent_t get_event() {
    if (ev_used == ev_pool.size()) {
      hipEvent_t e;
      SF_HIP(hipEventCreate(&e));
      ev_pool.push_back(e);
    }
    return ev_pool[ev_used++];
  }
  std::vector<std::pair<std::string, int64_t>> names;
  // graph cache for sf_vsample
  bool no_ln_fusion = tune_env("SF_NO_LN_FUSION") != nullptr;   // debugging aid: launch every LayerNorm separately
  bool no_indep_branches = tune_env("SF_NO_INDEP_BRANCHES") != nullptr;   // debugging aid: fork / join the branches inside every step
  bool no_thin_tail = tune_env("SF_NO_THIN_TAIL") != nullptr;   // debugging aid: conv2 / inject of the thin levels as two launches
  hipGraphExec_t gexec = nullptr;
  hipGraphExec_t gexec_br[8] = {};   // independent-branch pipelines: one single-stream step graph per branch
  bool gexec_indep = false;
  // the instantiated step graph is reused by later sf_vsample calls with the same shape / workspace / guidance scale
  // (every pointer baked into its kernel nodes lives in the workspace or in the engine; the sampler state is an
  // internal workspace buffer, the caller's x is copied in and out)
  struct GraphKey {
    int B = 0, L0 = 0, T = 0, nbr = 0;
    bool two = false, valid = false;
    float scale = 0.f;
    const void *ws = nullptr;
    bool operator==(const GraphKey &o) const {
      return valid && o.valid && B == o.B && L0 == o.L0 && T == o.T && nbr == o.nbr && two == o.two && scale == o.scale && ws == o.ws;
    }
  } gkey;
  // Step graphs of OTHER recently used (shape, workspace, guidance) combinations: a server alternating between a few request
  // shapes replays cached graphs instead of re-capturing and re-instantiating on every switch.
  struct GraphEntry {
    GraphKey key;
    hipGraphExec_t g = nullptr, br[8] = {};
    bool indep = false;
    void destroy() {
      if (g) (void)hipGraphExecDestroy(g);
      for (hipGraphExec_t &b : br)
        if (b) (void)hipGraphExecDestroy(b);
      g = nullptr;
      for (hipGraphExec_t &b : br) b = nullptr;
    }
  };
  static constexpr size_t kGraphStash = 3;
  std::vector<GraphEntry> gstash;   // most recently stashed last
  // make `key` the active graph set: a stashed set is swapped in, the displaced active set is stashed (oldest evicted)
  void activate_graphs(const GraphKey &key) {
    if (gkey == key) return;
    GraphEntry cur;
    cur.key = gkey;
    cur.g = gexec;
    cur.indep = gexec_indep;
    for (int i = 0; i < 8; ++i) cur.br[i] = gexec_br[i];
    gexec = nullptr;
    gexec_indep = false;
    for (hipGraphExec_t &b : gexec_br) b = nullptr;
    gkey.valid = false;
    for (size_t i = 0; i < gstash.size(); ++i)
      if (gstash[i].key == key) {
        gkey = gstash[i].key;
        gexec = gstash[i].g;
        gexec_indep = gstash[i].indep;
        for (int j = 0; j < 8; ++j) gexec_br[j] = gstash[i].br[j];
        gstash.erase(gstash.begin() + i);
        break;
      }
    if (cur.key.valid && (cur.g || cur.br[0])) {
      if (gstash.size() >= kGraphStash) {
        gstash.front().destroy();
        gstash.erase(gstash.begin());
      }
      gstash.push_back(cur);
    } else {
      cur.destroy();
    }
  }
  void drop_all_graphs() {
    gkey.valid = false;
    gexec_indep = false;
    if (gexec) (void)hipGraphExecDestroy(gexec);
    gexec = nullptr;
    for (hipGraphExec_t &g : gexec_br) {
      if (g) (void)hipGraphExecDestroy(g);
      g = nullptr;
    }
    for (GraphEntry &e : gstash) e.destroy();
    gstash.clear();
  }

  // stream capture is illegal on the legacy null stream (torch's default): the sampling loop runs on an
  // engine-owned stream, fenced against the caller's stream with events on both sides
  hipStream_t own_stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  // Clip-parallel branches: at small batch every kernel of the chain is latency-bound (a few hundred short
  // workgroups), so independent slices of the batch run the whole U-Net concurrently on separate streams
  // (fork/join with events; captured into the step graph as parallel branches).
  static constexpr int kMaxBranches = 8;
  hipStream_t bstream[kMaxBranches] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxBranches] = {};
  int branches_override = 0;
  // Tuning hook (tuning build only) SF_BRANCH_CUMASK=d: the stream of branch b is created with a CU mask that holds the mask bits k with
  // (k / d) % branches == b (hipExtStreamCreateWithCUMask) -- the round-6 A/B of CU-partitioned clip-parallel branches
  // (profiles/r6_d_xcd_branches.txt; tools/r6_probes.hip shows what a mask bit is on this machine).  0 / unset: plain streams.
  static hipError_t make_branch_stream(hipStream_t *st, int branch, int nbr) {
    static const int div = [] { const char *e = tune_env("SF_BRANCH_CUMASK"); return e ? atoi(e) : 0; }();
    if (div <= 0 || nbr < 2) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mask(words, 0u);
    for (int k = 0; k < ncu; ++k)
      if ((k / div) % nbr == branch) mask[k / 32] |= 1u << (k % 32);
    return hipExtStreamCreateWithCUMask(st, (uint32_t)words, mask.data());
  }
  int stream_nbr = 0;   // branch count the masked streams were created for
  void ensure_branch_streams(int n) {
    if (!ev_fork) SF_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (int i = 1; i < n; ++i) {
      if (!bstream[i]) SF_HIP(make_branch_stream(&bstream[i], i, n));
      if (!ev_join[i]) SF_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
    }

  }

  ~sf_unet() {
    drop_all_graphs();
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (int i = 0; i < kMaxBranches; ++i) {
      if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
      if (bstream[i]) (void)hipStreamDestroy(bstream[i]);
    }
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
};

namespace {

// ---------------------------------------------------------------------------------------------------
// construction: name lookup + packing
// ---------------------------------------------------------------------------------------------------
struct Builder {
  sf_unet &u;
  const WeightMap *wm;
  hipStream_t s;

  const float *get(const std::string &name, int64_t numel) {
    if (u.listing) {
      u.names.push_back({name, numel});
      return nullptr;
    }
    return wm->get(name, numel);
  }
  float *copy_f32(const std::string &name, int64_t numel) {
    const float *src = get(name, numel);
    if (u.listing) return nullptr;
    float *dst = u.arena.alloc_n<float>(numel);
    SF_HIP(hipMemcpyAsync(dst, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
    return dst;
  }

  // Conv1d weight (N, C1 + C2, taps); channels [0,C1) -> taps x cin_pad block, [C1, C1+C2) -> cin2_pad block (taps == 1)
  ConvW conv(const std::string &pre, int N, int C1, int taps, int C2, bool bias, bool direct, int cin_pad, int cin2_pad, bool cb = false) {
    ConvW c;
    c.N = N;
    c.taps = taps;
    c.direct = direct;
    c.cin = direct ? C1 : cin_pad;
    c.cin2 = C2 ? (direct ? C2 : cin2_pad) : 0;
    c.K = taps * c.cin + c.cin2;
    c.kreal = taps * C1 + C2;
    const int Ctot = C1 + C2;
    const float *w = get(pre + ".weight", (int64_t)N * Ctot * taps);
    const float *b = bias ? get(pre + ".bias", N) : nullptr;
    if (u.listing) return c;
    const int wdt = direct ? F32 : u.dt;
    c.w = u.arena.alloc((int64_t)N * c.K * dsize(wdt));
    SF_HIP(launch_pack_conv(wdt, w, N, Ctot, 0, C1, taps, c.cin, nullptr, c.w, c.K, 0, s));
    if (C2) SF_HIP(launch_pack_conv(wdt, w, N, Ctot, C1, C2, 1, c.cin2, nullptr, c.w, c.K, (int64_t)taps * c.cin, s));
    if (direct && N % 8 == 0 && C1 % 8 == 0) {   // 8-channel levels can also run on conv_thin (MFMA, compute type):
      const int c2p = pad_to(C2, 8);                // second source padded to whole octets (the context buffer is, too)
      const int64_t kt = (int64_t)taps * C1 + c2p;
      c.wt = u.arena.alloc(N * kt * dsize(u.dt));
      SF_HIP(launch_pack_conv(u.dt, w, N, Ctot, 0, C1, taps, C1, nullptr, c.wt, kt, 0, s));
      if (C2) SF_HIP(launch_pack_conv(u.dt, w, N, Ctot, C1, C2, 1, c2p, nullptr, c.wt, kt, (int64_t)taps * C1, s));
    }
    static const bool no_cb = tune_env("SF_NO_CB") != nullptr;   // debugging / A-B aid: keep the wave-private GEMM chain everywhere
    const int cbdt = u.x3 ? (int)F32X : u.dt;   // fp32x: (hi, lo') fp16 fragment pairs
    if (cb && !no_cb && !direct && taps == 3 && C2 == 0 && cin_pad == C1 && conv_cb_shape_ok(cbdt, 1, 64, C1, N, 0)) {
      // second copy in MFMA fragment order for the channel-block split-K kernel (the small-batch engine of the deep levels)
      c.wcb = u.arena.alloc((int64_t)conv_cb_weight_elems(N, C1) * dsize(cbdt));
      SF_HIP(launch_pack_conv_cb(cbdt, w, N, C1, c.wcb, s));
    }
    pack_wfr(c);
    if (b) {
      c.bias = u.arena.alloc_n<float>(N);
      SF_HIP(hipMemcpyAsync(c.bias, b, N * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return c;
  }

  // second copy of a packed [N][K] matrix in MFMA fragment order for the register-staged small-batch GEMM (conv_gemm_rs.hip)
  void pack_wfr(ConvW &c) {
    static const bool off = tune_env("SF_NO_RS") != nullptr;
    if (off || u.listing || c.direct || u.dt == F32 || !c.w || (c.K % 64) || c.K > 2048 || (c.N % 32) || (c.cin % 16) || (c.cin2 % 16)) return;
    c.wfr = u.arena.alloc((int64_t)c.N * c.K * dsize(u.dt));
    SF_HIP(launch_pack_wfr(u.dt, c.w, c.N, c.K, c.wfr, s));
  }

  // Linear weight (N, K) [+ per-column scale] -> [N][Kpad] in the compute type, rows appended at row0 of dst
  void linear_into(ConvW &dst, int row0, const float *w, int N, int K, const float *cscale) {
    if (u.listing) return;
    char *p = static_cast<char *>(dst.w) + (int64_t)row0 * dst.K * dsize(u.dt);
    SF_HIP(launch_pack_rows(u.dt, w, N, K, K, cscale, p, dst.K, s));
  }
  ConvW linear_alloc(int N, int Kpad, bool bias) {
    ConvW c;
    c.N = N;
    c.K = Kpad;
    c.cin = Kpad;
    c.taps = 1;
    if (u.listing) return c;
    c.w = u.arena.alloc((int64_t)N * Kpad * dsize(u.dt));
    SF_HIP(hipMemsetAsync(c.w, 0, (int64_t)N * Kpad * dsize(u.dt), s));
    if (bias) {
      c.bias = u.arena.alloc_n<float>(N);
      SF_HIP(hipMemsetAsync(c.bias, 0, N * sizeof(float), s));
    }
    return c;
  }
  ConvW linear(const std::string &pre, int N, int K, int Kpad, bool bias) {
    const float *w = get(pre + ".weight", (int64_t)N * K);
    const float *b = bias ? get(pre + ".bias", N) : nullptr;
    ConvW c = linear_alloc(N, Kpad, bias);
    c.kreal = K;
    linear_into(c, 0, w, N, K, nullptr);
    if (b && !u.listing) SF_HIP(hipMemcpyAsync(c.bias, b, N * sizeof(float), hipMemcpyDeviceToDevice, s));
    return c;
  }
};

void build_group(Builder &bd, Group &g, const std::string &pre, int d, int &mod_cols, int &ca_cols, int &n_ca) {
  sf_unet &u = bd.u;
  const sf_unet_config &c = u.cfg;
  const int C = c.channels[d];
  const bool thin = (C % 32) != 0;
  g.gn1_g = bd.copy_f32(pre + ".resnet.gn1.weight", C);
  g.gn1_b = bd.copy_f32(pre + ".resnet.gn1.bias", C);
  g.conv1 = bd.conv(pre + ".resnet.conv1", C, C, 3, 0, true, thin, C, 0, /*cb=*/true);
  g.gn2_g = bd.copy_f32(pre + ".resnet.gn2.weight", C);
  g.gn2_b = bd.copy_f32(pre + ".resnet.gn2.bias", C);
  g.conv2 = bd.conv(pre + ".resnet.conv2", C, C, 3, 0, true, thin, C, 0, /*cb=*/true);
  // Modulation Linear -> rows of the shared per-step GEMM (filled by build())
  g.mod_off = mod_cols;
  mod_cols = pad_to(mod_cols + 2 * C, 4);
  const int ctx = c.context_channels[d];
  g.inject = bd.conv(pre + ".inject.conv", C, C, 1, ctx, true, thin, C, pad_to(ctx, 32));
  g.attn = c.attentions[d] != 0;
  g.cross = c.cross_attentions[d] != 0;
  const int hd = u.hd;
  if (g.attn) {
    if (thin) fail(SF_ERR_UNSUPPORTED, "self-attention at depth %d needs channels %% 32 == 0 (got %d)", d, C);
    const std::string a = pre + ".attn";
    const float *ng = bd.get(a + ".norm.weight", C), *nb = bd.get(a + ".norm.bias", C);
    const float *cg = bd.get(a + ".norm_context.weight", C), *cb = bd.get(a + ".norm_context.bias", C);
    const float *wq = bd.get(a + ".to_q.weight", (int64_t)hd * C);
    const float *wkv = bd.get(a + ".to_kv.weight", (int64_t)2 * hd * C);
    const float *wo = bd.get(a + ".to_out.weight", (int64_t)C * hd);
    const float *bo = c.attention_out_bias ? bd.get(a + ".to_out.bias", C) : nullptr;
    g.qkv = bd.linear_alloc(3 * hd, C, true);
    g.attn_out = bd.linear_alloc(C, hd, c.attention_out_bias != 0);
    if (!u.listing) {
      // fold the LayerNorm affines:  W (g*xhat + b) = (W diag g) xhat + W b
      bd.linear_into(g.qkv, 0, wq, hd, C, ng);
      bd.linear_into(g.qkv, hd, wkv, 2 * hd, C, cg);
      SF_HIP(launch_fold_bias(wq, hd, C, nb, nullptr, g.qkv.bias, bd.s));
      SF_HIP(launch_fold_bias(wkv, 2 * hd, C, cb, nullptr, g.qkv.bias + hd, bd.s));
      bd.linear_into(g.attn_out, 0, wo, C, hd, nullptr);
      if (bo) SF_HIP(hipMemcpyAsync(g.attn_out.bias, bo, C * sizeof(float), hipMemcpyDeviceToDevice, bd.s));
      g.qkv_colsum = u.arena.alloc_n<float>(3 * hd);
      SF_HIP(launch_row_sums(u.dt, g.qkv.w, 3 * hd, g.qkv.K, g.qkv_colsum, bd.s));
      bd.pack_wfr(g.qkv);
      bd.pack_wfr(g.attn_out);
    }
  }
  if (g.cross) {
    // CrossAttentionItem over ONE context token: softmax == 1, so out = x + W_o (W_v LN(e)).
    // norm / to_q / the k half of to_kv cannot influence the result; they are still required parameters.
    const std::string a = pre + ".cross";
    const int E = c.embedding_features;
    bd.get(a + ".norm.weight", C);
    bd.get(a + ".norm.bias", C);
    bd.get(a + ".to_q.weight", (int64_t)hd * C);
    const float *cg = bd.get(a + ".norm_context.weight", E), *cb = bd.get(a + ".norm_context.bias", E);
    const float *wkv = bd.get(a + ".to_kv.weight", (int64_t)2 * hd * E);
    const float *wo = bd.get(a + ".to_out.weight", (int64_t)C * hd);
    const float *bo = c.attention_out_bias ? bd.get(a + ".to_out.bias", C) : nullptr;
    g.ca_idx = n_ca++;
    g.ca_off = ca_cols;
    ca_cols += C;
    g.cross_out = bd.linear_alloc(C, hd, c.attention_out_bias != 0);   // the bias rides in the collapsed per-clip vector
    if (!u.listing) {
      bd.linear_into(u.wv_cat, g.ca_idx * hd, wkv + (int64_t)hd * E, hd, E, cg);
      SF_HIP(launch_fold_bias(wkv + (int64_t)hd * E, hd, E, cb, nullptr, u.wv_cat.bias + g.ca_idx * hd, bd.s));
      bd.linear_into(g.cross_out, 0, wo, C, hd, nullptr);
      if (bo) SF_HIP(hipMemcpyAsync(g.cross_out.bias, bo, C * sizeof(float), hipMemcpyDeviceToDevice, bd.s));
    }
  }
}

void build(sf_unet &u, const WeightMap *wm, hipStream_t s) {
  const sf_unet_config &c = u.cfg;
  if (c.n_layers < 1 || c.n_layers > SF_MAX_DEPTH) fail(SF_ERR_INVALID, "n_layers out of range");
  if (c.attention_features != 64) fail(SF_ERR_UNSUPPORTED, "attention_features must be 64 (got %d)", c.attention_features);
  if (c.embedding_max_length != 1) fail(SF_ERR_UNSUPPORTED, "embedding_max_length must be 1 (CLAP embedding), got %d", c.embedding_max_length);
  if (c.modulation_features % 32 || c.embedding_features % 32) fail(SF_ERR_UNSUPPORTED, "modulation/embedding features must be multiples of 32");
  if (c.dtype != SF_F32 && c.dtype != SF_BF16 && c.dtype != SF_F16 && c.dtype != SF_F32X) fail(SF_ERR_INVALID, "bad dtype");
  u.x3 = c.dtype == SF_F32X;
  u.dt = u.x3 ? (int)SF_F32 : c.dtype;
  u.mf = c.modulation_features;
  u.hd = c.attention_heads * c.attention_features;
  if (c.time_fourier_features < 0 || c.time_fourier_features > 4096) fail(SF_ERR_INVALID, "time_fourier_features out of range");
  u.half = c.time_fourier_features > 0 ? c.time_fourier_features : u.mf / 2;
  u.four_ld = pad_to(1 + 2 * u.half, 32);
  Builder bd{u, wm, s};

  int n_cross = 0;
  for (int d = 0; d < c.n_layers; ++d) {
    if (c.context_channels[d] <= 0) fail(SF_ERR_UNSUPPORTED, "context_channels[%d] must be > 0", d);
    if (c.channels[d] % c.resnet_groups) fail(SF_ERR_INVALID, "channels[%d] not divisible by resnet_groups", d);
    if (c.channels[d] % 8) fail(SF_ERR_UNSUPPORTED, "channels[%d] must be a multiple of 8", d);
    int f = c.factors[d];
    if (f < 1 || (f & (f - 1))) fail(SF_ERR_UNSUPPORTED, "factors[%d] must be a power of two", d);
    if (c.cross_attentions[d]) n_cross += 2 * c.items[d];
  }
  u.wv_cat = bd.linear_alloc(n_cross > 0 ? n_cross * u.hd : 1, c.embedding_features, true);

  u.fourier_w = bd.copy_f32("net.time.fourier_w", u.half);
  u.lin0 = bd.linear("net.time.lin0", u.mf, 1 + 2 * u.half, u.four_ld, true);
  u.mlp0 = bd.linear("net.time.mlp.0", u.mf, u.mf, u.mf, true);
  u.mlp1 = bd.linear("net.time.mlp.1", u.mf, u.mf, u.mf, true);
  u.fixed_emb = bd.copy_f32("net.cfg.fixed_embedding.weight", (int64_t)c.embedding_max_length * c.embedding_features);

  int mod_cols = 0, ca_cols = 0, n_ca = 0;
  u.blocks.assign(c.n_layers, Block());
  int cin = c.in_channels;
  for (int d = 0; d < c.n_layers; ++d) {
    Block &b = u.blocks[d];
    b.C = c.channels[d];
    b.cin = cin;
    b.factor = c.factors[d];
    b.ctx = c.context_channels[d];
    while ((1 << b.up_shift) < b.factor) ++b.up_shift;
    const bool thin = (b.C % 32) != 0;
    // (pad columns are zero-filled by the layout conversion)
    b.ctx_ld = thin ? pad_to(b.ctx, 8) : pad_to(b.ctx, 32);
    const std::string pre = "net.blocks." + std::to_string(d);
    // Downsample: Conv1d(cin, C, kernel=f, stride=f).  As a GEMM it is a plain matrix product over the
    // (rows/f, f*cin) view of the input when that width is MFMA-friendly; else the direct kernel.
    const bool down_direct = ((b.factor * cin) % 32) != 0;
    if (down_direct && b.C > 32) fail(SF_ERR_UNSUPPORTED, "down conv at depth %d: %d -> %d needs (factor*in) %% 32 == 0", d, cin, b.C);
    b.down = bd.conv(pre + ".down", b.C, cin, b.factor, 0, true, down_direct, cin, 0);
    if (thin && cin > 32) fail(SF_ERR_UNSUPPORTED, "up conv at depth %d unsupported (%d -> %d)", d, b.C, cin);
    b.up_transposed = c.upsample_mode == SF_UP_TRANSPOSE;
    if (b.up_transposed) {
      // Upsample (a-unet `Upsample`): ConvTranspose1d(C -> cin, kernel = stride = f), weight (C, cin, f).  out[b, o, l*f + t] =
      // sum_c x[b, c, l] w[c][o][t] + bias[o]: the mirror image of the patchify down-conv, ONE GEMM with N = f * cin columns
      // over the (rows, f * cin) view of the output; bias and SkipModulate scale are tiled f times.
      const int N = b.factor * cin;
      const float *w = bd.get(pre + ".up.weight", (int64_t)b.C * cin * b.factor);
      const float *bi = bd.get(pre + ".up.bias", cin);
      ConvW &u_ = b.up;
      u_.N = N;
      u_.taps = 1;
      u_.direct = thin;
      u_.cin = b.C;
      u_.K = u_.kreal = b.C;
      if (!u.listing) {
        const int wdt = thin ? F32 : u.dt;
        u_.w = u.arena.alloc((int64_t)N * u_.K * dsize(wdt));
        SF_HIP(launch_pack_convT(wdt, w, b.C, cin, b.factor, u_.w, u_.K, s));
        u_.bias = u.arena.alloc_n<float>(N);
        for (int t = 0; t < b.factor; ++t) SF_HIP(hipMemcpyAsync(u_.bias + (int64_t)t * cin, bi, cin * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
    } else {
      // UpsampleInterpolate: nearest x f then Conv1d(C, cin, 3, padding=1)
      b.up = bd.conv(pre + ".up", cin, b.C, 3, 0, true, thin, b.C, 0);
    }
    b.skip_off = mod_cols;
    b.skip_cols = b.up_transposed ? b.factor * cin : cin;
    mod_cols = pad_to(mod_cols + b.skip_cols, 4);   // every entry of the shared modulation vector starts 16-byte aligned
    b.down_items.assign(c.items[d], Group());
    b.up_items.assign(c.items[d], Group());
    for (int j = 0; j < c.items[d]; ++j) build_group(bd, b.down_items[j], pre + ".items_down." + std::to_string(j), d, mod_cols, ca_cols, n_ca);
    for (int j = 0; j < c.items[d]; ++j) build_group(bd, b.up_items[j], pre + ".items_up." + std::to_string(j), d, mod_cols, ca_cols, n_ca);
    cin = b.C;
  }
  u.mod_cols = mod_cols;
  u.mod_ld = pad_to(mod_cols, 4);
  u.ca_cols = ca_cols;
  u.ca_ld = pad_to(ca_cols > 0 ? ca_cols : 1, 4);
  u.n_ca = n_ca;

  // All Modulation / SkipModulate Linear layers share their input SiLU(features): one GEMM per step.
  u.mod = bd.linear_alloc(mod_cols, u.mf, true);
  for (int d = 0; d < c.n_layers; ++d) {
    Block &b = u.blocks[d];
    const std::string pre = "net.blocks." + std::to_string(d);
    {
      const float *w = bd.get(pre + ".skip.to_scale.weight", (int64_t)b.cin * u.mf);
      const float *bi = bd.get(pre + ".skip.to_scale.bias", b.cin);
      if (!u.listing) {
        for (int t = 0; t < b.skip_cols / b.cin; ++t) {
          bd.linear_into(u.mod, b.skip_off + t * b.cin, w, b.cin, u.mf, nullptr);
          SF_HIP(hipMemcpyAsync(u.mod.bias + b.skip_off + t * b.cin, bi, b.cin * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
      }
    }
    for (int pass = 0; pass < 2; ++pass) {
      auto &items = pass == 0 ? b.down_items : b.up_items;
      for (size_t j = 0; j < items.size(); ++j) {
        const std::string gp = pre + (pass == 0 ? ".items_down." : ".items_up.") + std::to_string(j) + ".mod.to_scale_shift";
        const float *w = bd.get(gp + ".weight", (int64_t)2 * b.C * u.mf);
        const float *bi = bd.get(gp + ".bias", 2 * b.C);
        if (!u.listing) {
          bd.linear_into(u.mod, items[j].mod_off, w, 2 * b.C, u.mf, nullptr);
          SF_HIP(hipMemcpyAsync(u.mod.bias + items[j].mod_off, bi, 2 * b.C * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
      }
    }
  }
  if (!u.listing && u.n_ca > 0 && !tune_env("SF_NO_CA_GROUPED")) {   // table of the per-item projections (conditioning() runs them as one launch)
    std::vector<CrossOutItem> items;
    std::vector<int2> blks;
    for (int d = 0; d < c.n_layers; ++d)
      for (int pass = 0; pass < 2; ++pass)
        for (const Group &g : (pass == 0 ? u.blocks[d].down_items : u.blocks[d].up_items)) {
          if (!g.cross || g.cross_out.direct || !g.cross_out.w) continue;
          CrossOutItem it;
          it.w = g.cross_out.w;
          it.bias = g.cross_out.bias;
          it.N = g.cross_out.N;
          it.ldw = g.cross_out.K;
          it.v_off = g.ca_idx * u.hd;
          it.out_off = g.ca_off;
          for (int c0 = 0; c0 < it.N; c0 += 64) blks.push_back(make_int2((int)items.size(), c0));
          items.push_back(it);
        }
    if ((int)items.size() == u.n_ca && u.hd <= 1024 && (u.hd % 32) == 0) {
      u.ca_items = static_cast<CrossOutItem *>(u.arena.alloc((int64_t)items.size() * sizeof(CrossOutItem)));
      u.ca_blocks = static_cast<int2 *>(u.arena.alloc((int64_t)blks.size() * sizeof(int2)));
      SF_HIP(hipMemcpyAsync(u.ca_items, items.data(), items.size() * sizeof(CrossOutItem), hipMemcpyHostToDevice, s));
      SF_HIP(hipMemcpyAsync(u.ca_blocks, blks.data(), blks.size() * sizeof(int2), hipMemcpyHostToDevice, s));
      SF_HIP(hipStreamSynchronize(s));   // the host vectors die here
      u.ca_nblocks = (int)blks.size();
    }
  }
  if (!u.listing && u.x3) {
    // fp32x: every MFMA-path matrix a second time as split fp16 operands (after all linear_into() fills and affine folds)
    auto split = [&](ConvW &w) {
      if (w.direct || !w.w || (w.K % 32)) return;   // (the kernels check the channel counts of the launch they are given)
      w.wx = u.arena.alloc((int64_t)w.N * w.K * 4);
      SF_HIP(launch_pack_wx(static_cast<const float *>(w.w), w.N, w.K, w.wx, s));
      // ... and in fragment order for the register-staged small-batch GEMM (InjectChannels, attention projections, down / up convolutions)
      if ((w.K % 64) == 0 && w.K <= 1280 && (w.N % 32) == 0 && (w.cin % 16) == 0 && (w.cin2 % 16) == 0) {
        w.wfrx = u.arena.alloc((int64_t)w.N * w.K * 4);
        SF_HIP(launch_pack_wfrx(static_cast<const float *>(w.w), w.N, w.K, w.wfrx, s));
      }
    };
    for (ConvW *w : {&u.lin0, &u.mlp0, &u.mlp1, &u.mod, &u.wv_cat}) split(*w);
    for (Block &b : u.blocks) {
      split(b.down);
      split(b.up);
      for (int pass = 0; pass < 2; ++pass)
        for (Group &g : (pass == 0 ? b.down_items : b.up_items))
          for (ConvW *w : {&g.conv1, &g.conv2, &g.inject, &g.qkv, &g.attn_out, &g.cross_out}) split(*w);
    }
  }
  if (!u.listing) SF_HIP(hipStreamSynchronize(s));
}

// ---------------------------------------------------------------------------------------------------
// workspace plan
// ---------------------------------------------------------------------------------------------------
constexpr int kSchedSteps = 1024;   // schedule capacity that keeps the workspace layout independent of num_steps

Plan make_plan(const sf_unet &u, Workspace &ws, int B, int L0, bool two, int num_steps) {
  const sf_unet_config &c = u.cfg;
  Plan p;
  p.B = B;
  p.Bt = two ? 2 * B : B;
  p.L0 = L0;
  p.two = two;
  int64_t stride = 1;
  for (int d = 0; d < c.n_layers; ++d) stride *= c.factors[d];
  if (B < 1 || L0 < 1 || L0 % stride) fail(SF_ERR_SHAPE, "L0=%d must be a positive multiple of the U-Net stride %lld", L0, (long long)stride);
  const size_t es = dsize(u.dt);
  p.lv.resize(c.n_layers);
  int L = L0;
  int64_t slab_floats = 16;
  for (int d = 0; d < c.n_layers; ++d) {
    Level &l = p.lv[d];
    L /= c.factors[d];
    l.L = L;
    l.C = c.channels[d];
    l.rows = (int64_t)p.Bt * L;
    for (int i = 0; i < 3; ++i) l.buf[i] = ws.alloc(l.rows * l.C * es);
    l.act = ws.alloc(l.rows * l.C * es);
    if (c.attentions[d]) {
      l.qkv = ws.alloc(l.rows * 3 * u.hd * es);
      l.ao = ws.alloc(l.rows * u.hd * es);
    }
    l.ctx = ws.alloc(l.rows * u.blocks[d].ctx_ld * es);
  }
  // branches: independent slices of the (doubled) batch; each gets its own GroupNorm scratch
  {
    // measured (bf16, MI355X): 8 evaluations per step: 2 branches 455 vs 1 branch 372 steps/s (more branches do not help: the
    // chain length, not the rows per launch, sets the time); 16: 320 vs 262; from 32 evaluations on the launches are long enough
    // to fill the chip and ONE branch wins (32: 210 vs 203, 32 with guidance 211 vs 198, 64 with guidance 154 vs 151) -- it also
    // spares the two-stream step graph that guidance needs
    // (re-measured with the later kernels: 32 evaluations 232 -> 253 steps/s with two branches, 16 clips with guidance 233 -> 243,
    // 40 / 48 evaluations +1 ... 3 %, 64 evaluations 163 -> 153: two branches up to 48 evaluations)
    // (round 3, after the hosted weight prefetch and the 128x64 rule for under-filled macro-tile launches: 64 evaluations -- batch 32
    // with guidance, BASELINE configs[2] -- 168.2 -> 182.0 steps/s with two branches; 96 evaluations 113.7 -> 131.6; 128 evaluations
    // 101.5 -> 108.3; four branches at 64 evaluations 166: profiles/r3_g_ab_branches.txt)
    static const int two_max = [] {   // tuning hook: largest number of evaluations per step that still runs as two branches
      const char *e = tune_env("SF_TWO_BRANCH_MAX");
      return e ? atoi(e) : 128;
    }();
    int want = u.branches_override > 0 ? u.branches_override : ((p.Bt >= 4 && p.Bt <= two_max) ? 2 : 1);
    if (want > sf_unet::kMaxBranches) want = sf_unet::kMaxBranches;
    while (want > 1 && p.Bt % want) --want;
    p.nbr = u.dbg.buf ? 1 : want;
    p.nbr_total = p.nbr;
    // GroupNorm partial-statistics scratch per branch: two slabs (statistics of x and of the hidden activation),
    // each [clips][chunks][groups][2]; the thin levels chunk by their workgroup tile (conv_thin_plan)
    int64_t one = (int64_t)(p.Bt / p.nbr) * 32 * c.resnet_groups * 2;
    for (int d = 0; d < c.n_layers; ++d) {
      const ThinPlan tp = conv_thin_plan(p.Bt / p.nbr, p.lv[d].L, p.lv[d].C);
      one = std::max<int64_t>(one, (int64_t)(p.Bt / p.nbr) * tp.nchw * c.resnet_groups * 2);
    }
    p.slab_half = align_up(one, 64);
    p.slab_stride = 2 * p.slab_half;
    slab_floats = p.slab_stride * p.nbr;
  }
  const int64_t n0 = (int64_t)p.Bt * L0 * c.in_channels;
  p.x2 = ws.alloc_n<float>(n0);
  p.vout = ws.alloc_n<float>(n0);
  p.xs = ws.alloc_n<float>(n0);
  p.mod_all = ws.alloc_n<float>((int64_t)p.Bt * u.mod_ld);
  p.mod_stride = u.mod_ld;
  p.steps_cap = num_steps > 1 ? num_steps : 0;
  p.ca_all = ws.alloc_n<float>((int64_t)p.Bt * u.ca_ld);
  p.slab = ws.alloc_n<float>(slab_floats);
  {
    int64_t one = 64;
    for (int d = 0; d < c.n_layers; ++d)
      if (p.lv[d].C % 32 == 0) one = std::max<int64_t>(one, p.lv[d].rows / p.nbr * (p.lv[d].C / 32) * 2);
    p.rowpart_half = align_up(one, 64);
    p.rowpart_stride = 2 * p.rowpart_half;
    p.rowpart = ws.alloc_n<float>(p.rowpart_stride * p.nbr);
  }
  {
    // Channel-block split-K chain (conv_cb.hip) for the levels whose per-branch activations are short: there the wave-private GEMMs
    // stream 4-12x their operands through single CUs and every launch is latency-bound.  The price is the fp32 partial slabs,
    // S * rows * C * 4 bytes written and read back per convolution: 5.8 MB per level at four clips per branch, 46 MB at 32 -- where
    // the step is bandwidth- / MFMA-bound and the slab traffic costs more than the operand amplification it removes (configs[2]
    // 191 vs 200 steps/s, one GPU's share of configs[3] 271 vs 285 with the chain on at 704-1408 rows: profiles/r4_d_ab_cb_rows.txt).
    // Footprint limit swept at batch 12 / 16 / 32 with and without guidance (profiles/r4_d_ab_cb_slab.txt): 12 MB is never behind "off"
    // beyond run-to-run noise and +1.5 % at batch 12-16.
    static const int cb_max_rows = [] {   // tuning hook: most rows per branch a level may have and still take the chain (0: never)
      const char *e = tune_env("SF_CB_MAX_ROWS");
      return e ? atoi(e) : 1408;
    }();
    static const double cb_max_slab_mb = [] {   // tuning hook: largest partial-slab footprint (MB per branch and level)
      const char *e = tune_env("SF_CB_MAX_SLAB_MB");
      return e ? atof(e) : 12.0;
    }();
    static const int cb_min_c = [] {      // tuning hook: narrowest level that takes the chain
      const char *e = tune_env("SF_CB_MIN_C");
      return e ? atoi(e) : 256;
    }();
    int64_t need = 0, need_gp = 0;
    for (int d = 0; d < c.n_layers; ++d) {
      Level &l = p.lv[d];
      const int bt = p.Bt / p.nbr;
      const int64_t rows = (int64_t)bt * l.L;
      const Block &b = u.blocks[d];
      const bool have_w = !b.down_items.empty() && b.down_items[0].conv1.wcb != nullptr;
      const bool shape_ok = have_w && l.C >= cb_min_c && rows <= cb_max_rows && conv_cb_shape_ok(u.x3 ? (int)F32X : u.dt, bt, l.L, l.C, l.C, c.resnet_groups) &&
                            cb_gn_plan(l.L).nch <= 32 && (int64_t)bt * 32 * c.resnet_groups * 2 <= p.slab_half;
      const double slab1 = (double)(l.C / 128) * rows * l.C * 4.0;   // bytes of the partial slabs with one channel block per workgroup
      static const bool no_kb2 = tune_env("SF_CB_NO_KB2") != nullptr;   // A/B aid
      static const double cb_max_slab2_mb = [] {   // tuning hook: the same limit for the two-block form
        const char *e = tune_env("SF_CB_MAX_SLAB2_MB");
        return e ? atof(e) : 12.0;
      }();
      l.cb = false;
      l.cb_kb = 1;
      if (shape_ok && slab1 <= cb_max_slab_mb * 1048576.0) l.cb = true;
      else if (shape_ok && !u.x3 && !no_kb2 && (l.C / 128) % 2 == 0 && l.C / c.resnet_groups >= 32 && slab1 * 0.5 <= cb_max_slab2_mb * 1048576.0) {
        l.cb = true;   // two channel blocks per workgroup: half the slabs (8-16 clips per branch at the 1024-channel levels)
        l.cb_kb = 2;
      }
      if (l.cb) need = std::max<int64_t>(need, (int64_t)(l.C / 128 / l.cb_kb) * rows * l.C);
      if (l.cb) need_gp = std::max<int64_t>(need_gp, ((rows + 31) / 32) * (l.C / 32) * 4);
    }
    p.cbslab_stride = align_up(need, 64);
    p.cbslab = need ? ws.alloc_n<float>(p.cbslab_stride * p.nbr) : nullptr;
    p.gnpart_stride = align_up(need_gp, 64);
    p.gnpart = need_gp ? ws.alloc_n<float>(p.gnpart_stride * p.nbr) : nullptr;
  }
  p.emb2 = ws.alloc_n<float>((int64_t)p.Bt * c.embedding_features);
  p.emb_t = ws.alloc((int64_t)p.Bt * c.embedding_features * es);
  p.xhat_e = ws.alloc((int64_t)p.Bt * c.embedding_features * es);
  p.v_all = ws.alloc((int64_t)p.Bt * (u.n_ca > 0 ? u.n_ca : 1) * u.hd * es);
  // Everything whose SIZE depends on the number of steps comes last, so that no other address does: a step graph
  // instantiated for one num_steps stays valid for another (up to kSchedSteps; the tables are indexed by the device
  // step counter).  Order: step counter, schedule (fixed capacity), sigmas, modulation table, time-MLP rows.
  const int64_t ns = std::max<int64_t>(num_steps > 0 ? num_steps : 1, kSchedSteps);
  p.step = ws.alloc_n<int>(16);   // device step counters: one per independent branch pipeline
  p.sched = ws.alloc_n<float>(4 * ns);
  p.sigs = ws.alloc_n<float>(ns + p.Bt);
  if (p.steps_cap) p.mod_steps = ws.alloc_n<float>((int64_t)p.steps_cap * u.mod_ld);   // base fixed, length ~ num_steps
  const int64_t frows = std::max<int64_t>(p.Bt, p.steps_cap);   // time-MLP rows: only used outside the step graph
  p.four = ws.alloc(frows * u.four_ld * es);
  p.f1 = ws.alloc(frows * u.mf * es);
  p.f2 = ws.alloc(frows * u.mf * es);
  p.sf = ws.alloc(frows * u.mf * es);
  return p;
}

// ---------------------------------------------------------------------------------------------------
// execution
// ---------------------------------------------------------------------------------------------------
struct Exec {
  sf_unet &u;
  Plan &p;
  hipStream_t s;
  const void *stats_of = nullptr;   // thin levels: the activation whose GroupNorm partials currently sit in p.slab
  const void *gnpart_of = nullptr;  // channel-block levels: the activation whose GroupNorm tile sums currently sit in p.gnpart
  bool up_gnpart = false;           // block(d + 1) left the tile sums of its output (level d's next item input) in p.gnpart
  int cur_depth = -1;               // depth tag of the launches being issued (profile records, roofline.depth_groups)

  template <class F> void timed(const char *label, double flops, double bytes, F &&f) {
#ifdef SF_TUNING_HOOKS   // (does not exist in the product library: a leaked variable would silently corrupt audio)
    {
      // measurement aid (phase removal): SF_SKIP_LABELS=gn_silu,ln_modulate,... drops every launch whose label starts with one of the
      // listed prefixes -- results are garbage, the step's wall time shows what that kernel family costs inside the two-branch step
      static const std::vector<std::string> skip = [] {
        std::vector<std::string> v;
        if (const char *e = tune_env("SF_SKIP_LABELS")) {
          std::string all(e);
          size_t a = 0;
          while (a <= all.size()) {
            const size_t b = all.find(',', a);
            const std::string t = all.substr(a, b == std::string::npos ? std::string::npos : b - a);
            if (!t.empty()) v.push_back(t);
            if (b == std::string::npos) break;
            a = b + 1;
          }
        }
        return v;
      }();
      for (const std::string &t : skip)
        if (strncmp(label, t.c_str(), t.size()) == 0) return;
    }
#endif
    ++u.launches;
    if (!u.prof_on) {
      f();
      return;
    }
    hipEvent_t e0 = u.get_event(), e1 = u.get_event();
    SF_HIP(hipEventRecord(e0, s));
    f();
    SF_HIP(hipEventRecord(e1, s));
    u.prof.push_back({label, flops, bytes, e0, e1, 0.f, cur_depth});
  }

  // Hosted weight prefetch (kernels.h, Prefetch): the launch that PRECEDES a GEMM carries extra workgroups that read the GEMM's weights,
  // so the GEMM streams them from the Infinity Cache instead of HBM (in the two-branch step: -1.0 us per GEMM launch, measured with a
  // separate touch launch in front of every GEMM, profiles/r3_d_touch_*).  `host_wgs` = workgroups of the hosting launch: the
  // prefetch takes the CUs it leaves idle.
  // `rows` = output rows of the GEMM that will read the weights: short activations run on the register-staged kernel, which reads
  // the fragment-ordered copy (conv_gemm_rs.hip)
  Prefetch pf_for(const ConvW &w, int host_wgs, int64_t rows = -1) const {
    const bool rs = (w.wfr || w.wfrx) && rows >= 0 && conv_gemm_rs_rows_ok(rows, w.N);
    return pf_bytes(w.direct ? nullptr : (rs ? (w.wfrx ? w.wfrx : w.wfr) : (w.wx ? w.wx : w.w)), (size_t)w.N * w.K * dsize(u.dt), host_wgs);
  }
  Prefetch pf_cb(const ConvW &w, int host_wgs) const { return pf_bytes(w.wcb, conv_cb_weight_elems(w.N, w.cin) * dsize(u.dt), host_wgs); }
  Prefetch pf_bytes(const void *ptr, size_t bytes, int host_wgs) const {
    static const bool off = tune_env("SF_NO_PREFETCH") != nullptr;
    Prefetch pf;
    static const int host_max = [] {   // tuning hook: hosts with more workgroups than this lend nothing
      const char *e = tune_env("SF_PF_HOST_MAX");
      return e ? atoi(e) : 1024;   // (208 at first: batch 32 without guidance 277.2 -> 281.1 steps/s with 1024, configs[2] unchanged)
    }();
    if (off || !ptr || host_wgs > host_max) return pf;   // a host that fills the chip has no idle CUs to lend
    if (bytes < (64u << 10) || bytes > 0x7FFFFFF0ull) return pf;   // small matrices: nothing to gain
    pf.ptr = ptr;
    pf.bytes = (unsigned)bytes;
    static const int cap = [] {   // tuning hook: upper bound of prefetch workgroups per host launch
      const char *e = tune_env("SF_PF_WGS");
      return e && atoi(e) > 0 ? atoi(e) : 224;
    }();
    static const int mult = [] {
      const char *e = tune_env("SF_PF_MULT");
      return e && atoi(e) > 0 ? atoi(e) : 1;
    }();
    pf.wgs = std::max(std::min(48, cap), std::min(cap, mult * (256 - host_wgs)));
    return pf;
  }

  // the producer of an item input of level d should leave GroupNorm tile sums (ConvGemmArgs::gnpart_out) for conv_cb's prologue
  bool wants_gnpart(int d) const {
    static const bool off = tune_env("SF_NO_CB_TILESTATS") != nullptr;   // A/B aid: keep the gn_silu launch in front of conv1
    if (off || d < 0 || d >= (int)p.lv.size()) return false;
    const Level &l = p.lv[d];
    return l.cb && p.gnpart && conv_cb_tile_stats_ok(l.L, l.C, u.cfg.resnet_groups);
  }
  // arm `a` (a GEMM whose output is the next item's input at level d) when the kernel it will run on can write them
  bool arm_gnpart(const ConvW &w, ConvGemmArgs &a, int d) const {
    if (!wants_gnpart(d)) return false;
    a.gnpart_out = p.gnpart;
    if (conv_gemm_emits_gnpart(u.dt, filled(w, a))) return true;
    a.gnpart_out = nullptr;
    return false;
  }

  ConvGemmArgs filled(const ConvW &w, ConvGemmArgs a) const {
    a.w = w.w;
    a.wfr = w.wfr;
    a.wx = w.wx;
    a.wfrx = w.wfrx;
    a.solo = p.nbr_total <= 1 ? 1 : 0;
    a.bias = w.bias;
    a.N = w.N;
    a.K = w.K;
    a.cin = w.cin;
    a.cin2 = w.cin2;
    a.taps = w.taps;
    if (a.n_store == 0) a.n_store = w.N;
    return a;
  }

  // ln: the GEMM normalises (and modulates) its first source on the fly from the producer's row partials
  void conv(const ConvW &w, ConvGemmArgs a0, int dt_in, int dt_out, bool ln = false) {
    ConvGemmArgs a = filled(w, a0);
    // algorithmic work of this launch: 2*M*N*K_real FLOPs; bytes = activations in + out (+ residual) + weights
    const double kreal = w.kreal > 0 ? w.kreal : w.K;
    const double es_in = dsize(dt_in), es_out = a.out_f32 ? 4.0 : (double)dsize(dt_out);
    const double flops = 2.0 * a.M * w.N * kreal;
    const double src_rows = (double)a.M / a.Lout * a.Lsrc;
    const double bytes = src_rows * (kreal / (w.taps > 0 ? w.taps : 1)) * es_in + (double)a.M * w.N * es_out * (a.res ? 2 : 1) +
                         (double)w.N * kreal * (w.direct ? 4.0 : (double)dsize(u.dt));
    if (w.direct) timed("conv_direct", flops, bytes, [&] { SF_HIP(launch_conv_direct(dt_in, dt_out, a, s)); });
    else {
      // measurement aid: SF_TOUCH=n reads this GEMM's weights with an n-workgroup kernel right before it,
      // to price what a weight prefetch hosted by the preceding kernel's idle CUs could save inside the real two-branch step
      static const int touch = [] {
        const char *e = tune_env("SF_TOUCH");
        return e ? atoi(e) : 0;
      }();
      if (touch > 0 && !u.prof_on) SF_HIP(launch_touch(w.w, (size_t)w.N * w.K * dsize(u.dt), touch, reinterpret_cast<unsigned *>(p.step + 8), s));
      if (dt_in != u.dt || (dt_out != u.dt && !a.out_f32)) fail(SF_ERR_INVALID, "internal: dtype mismatch on the MFMA path");
      if (ln) timed(conv_gemm_ln_variant_name(u.dt, a), flops + 8.0 * a.M * a.cin, bytes,
                    [&] { SF_HIP(launch_conv_gemm_ln(u.dt, a, s)); });
      else timed(conv_gemm_variant_name(u.dt, a), flops, bytes, [&] { SF_HIP(launch_conv_gemm(u.dt, a, s)); });
    }
  }

  // rows x K GEMM on per-clip vectors (time MLP, modulation, cross-attention collapse)
  void dense(const ConvW &w, const void *src, int src_ld, int rows, void *out, int out_ld, int act, bool out_f32, int col0 = 0,
             int ncols = -1) {
    ConvGemmArgs a;
    a.src = src;
    a.src_ld = src_ld;
    a.M = rows;
    a.Lout = a.Lsrc = 1;
    a.out = out;
    a.out_ld = out_ld;
    a.act = act;
    a.out_f32 = out_f32 ? 1 : 0;
    (void)col0;
    (void)ncols;
    conv(w, a, u.dt, out_f32 ? F32 : u.dt);
  }

  void gn(const void *x, int d, int C) {
    const Level &l = p.lv[d];
    GnPlan gp = gn_plan(p.Bt, l.L, C);
    timed("gn_stats", 3.0 * l.rows * C, (double)l.rows * C * dsize(u.dt),
          [&] { SF_HIP(launch_gn_stats(u.dt, x, C, p.Bt, l.L, C, u.cfg.resnet_groups, gp.nch, gp.chunk_rows, p.slab, s)); });
  }

  // One item-group: Resnet -> Modulation -> InjectChannels -> [Attention] -> [CrossAttention]
  // cur is consumed; returns the buffer that holds the result.  tA / tB are the two free buffers.
  // next: the item group that consumes this one's output at the same level (nullptr: a down / up convolution follows)
  void group(const Group &g, int d, void *&cur, void *&tA, void *&tB, const std::string &tapname, const Group *next = nullptr) {
    const Level &l = p.lv[d];
    const Block &b = u.blocks[d];
    const int C = l.C, G = u.cfg.resnet_groups;
    // ResnetItem: x + Conv3(SiLU(GN(Conv3(SiLU(GN(x)))))).  GroupNorm+SiLU is materialised once per conv
    // (gn_silu) rather than applied in the GEMM's A-load: the activation would otherwise be recomputed for every
    // column tile and tap of the wide layers.
    // Thin levels (C <= 64: at most two column tiles) keep the activation in the conv's A-load instead
    // (statistics from gn_stats): their tensors are long and the extra activated copy would cost more.
    if (group_thin(g, d, cur, tA, tB)) {
      if (!g.attn) {
        u.dbg.tap(tapname, u.dt, cur, C, l.rows, C, s);
        return;
      }
    } else {
    const bool fuse_act = C <= 64;
    GnPlan gp = gn_plan(p.Bt, l.L, C);
    auto conv3 = [&](const ConvW &w, const void *in, void *out, const float *gam, const float *bet, const void *res, float *rowpart,
                     const ConvW *next = nullptr) {
      ConvGemmArgs a;
      if (next) a.pf = pf_for(*next, (int)((l.rows + 31) / 32) * ((C + 31) / 32), l.rows);
      if (rowpart) {
        a.rowpart_out = rowpart;
        a.rowpart_nt = C / 32;
      }
      if (fuse_act) {
        gn(in, d, C);
        a.src = in;
        a.pro = 1;
        a.G = G;
        a.nch = gp.nch;
        a.chunk_rows = gp.chunk_rows;
        a.stats = p.slab;
        a.gamma = gam;
        a.beta = bet;
        a.eps = 1e-5f;
      } else {
        a.src = l.act;
      }
      a.src_ld = C;
      a.M = (int)l.rows;
      a.Lout = a.Lsrc = l.L;
      a.stride = 1;
      a.pad = 1;
      a.out = out;
      a.out_ld = C;
      a.res = res;
      a.res_ld = C;
      if (!fuse_act) {
        // fp32x: when the convolution runs on the macro tiles, GroupNorm+SiLU writes its rows already split into fp16 (hi, lo') -- the
        // activated tensor has no other reader -- and the GEMM spends no vector instruction on its activation operand
        const bool xf = u.x3 && (C % 32) == 0 && conv_gemm_src_x3_ok(filled(w, a));
        // (the chunked two-launch form of launch_gn_silu_ws was measured on the 2^18-sample shape: 97.4 vs 98.4 steps/s -- the second
        // launch costs what the better CU fill saves -- so the engine keeps the single launch)
        timed("gn_silu", 12.0 * l.rows * C, 3.0 * l.rows * C * dsize(u.dt),
              [&] { SF_HIP(launch_gn_silu(u.dt, in, C, p.Bt, l.L, C, G, gam, bet, 1e-5f, l.act, C, s, pf_for(w, p.Bt * G), xf)); });
        a.src_x3 = xf ? 1 : 0;
      }
      conv(w, a, u.dt, u.dt);
    };
    // Wide levels: the two LayerNorms of an item (Modulation before InjectChannels, the attention pre-norm) are not
    // launched: the producing GEMM's epilogue leaves per-row (mean, M2) partials per 32-column tile, the consuming GEMM
    // pools them and normalises (and modulates) its A operand while staging it.  Falls back to ln_modulate launches
    // whenever a shape is outside what conv_gemm_fast / conv_gemm_wp cover.
    float *rp_y = p.rowpart, *rp_z = p.rowpart + p.rowpart_half;
    auto inject_args = [&](const void *src, void *out) {
      ConvGemmArgs a;
      a.src = src;
      a.src_ld = C;
      a.src2 = l.ctx;
      a.src2_ld = b.ctx_ld;
      a.M = (int)l.rows;
      a.Lout = a.Lsrc = l.L;
      a.out = out;
      a.out_ld = C;
      a.res = src;
      a.res_ld = C;
      if (g.cross && !g.attn) {
        a.badd = p.ca_all + g.ca_off;
        a.badd_ld = u.ca_ld;
      }
      return a;
    };
    auto qkv_args = [&](const void *src) {
      ConvGemmArgs a;
      a.src = src;
      a.src_ld = C;
      a.M = (int)l.rows;
      a.Lout = a.Lsrc = l.L;
      a.out = l.qkv;
      a.out_ld = 3 * u.hd;
      return a;
    };
    bool fuse_mod = false;
    // (measured: normalising the operand while staging costs more than the separate launch once the row is 1024 wide)
    static const int ln_fuse_maxc = [] {   // tuning hook: widest level whose Modulation LayerNorm is folded into the InjectChannels GEMM
      const char *e = tune_env("SF_LN_FUSE_MAXC");
      return e ? atoi(e) : 512;
    }();
    if (!fuse_act && C % 32 == 0 && C <= ln_fuse_maxc && !u.no_ln_fusion) {
      ConvGemmArgs pc;   // conv2 as it will be launched
      pc.src = l.act;
      pc.src_ld = C;
      pc.M = (int)l.rows;
      pc.Lout = pc.Lsrc = l.L;
      pc.pad = 1;
      pc.out = tB;
      pc.out_ld = C;
      pc.res = cur;
      pc.res_ld = C;
      ConvGemmArgs pi = inject_args(tB, tA);
      pi.ln_part = rp_y;
      pi.ln_nt = C / 32;
      pi.ln_ss = p.mod_all + g.mod_off;
      pi.ln_ss_ld = p.mod_stride;
      pi.ln_eps = 1e-6f;
      pi.res_ln = 1;
      fuse_mod = conv_gemm_emits_rowpart(u.dt, filled(g.conv2, pc)) && conv_gemm_ln_ok(u.dt, filled(g.inject, pi));
    }
    const bool use_cb = l.cb && !fuse_act && g.conv1.wcb && g.conv2.wcb && g.conv1.bias && g.conv2.bias && p.cbslab;
    bool fuse_attn = false;
    if (g.attn && C % 32 == 0 && !u.no_ln_fusion) {
      ConvGemmArgs pq = qkv_args(tB);
      pq.ln_part = rp_z;
      pq.ln_nt = C / 32;
      pq.ln_eps = 1e-5f;
      pq.ln_colsum = g.qkv_colsum;
      ConvGemmArgs pi = inject_args(tA, tB);
      // the LN-folded InjectChannels kernel always writes row partials; the channel-block chain launches the PLAIN InjectChannels GEMM
      // whatever fuse_mod says, so ask about the launch that will actually run
      const bool inj_emits = (fuse_mod && !use_cb) ? true : conv_gemm_emits_rowpart(u.dt, filled(g.inject, pi));
      fuse_attn = inj_emits && conv_gemm_ln_ok(u.dt, filled(g.qkv, pq));
    }
    if (use_cb) {
      // Channel-block split-K chain (conv_cb.hip): the two launches that followed the convolutions anyway (GroupNorm+SiLU, LayerNorm +
      // Modulation) sum the fp32 partial slabs; the second convolution applies GroupNorm+SiLU while staging its activation panel.
      const int bt = p.Bt, kb = l.cb_kb, S = C / 128 / kb;
      const double es = dsize(u.dt), rc = (double)l.rows * C;
      const double cflops = 2.0 * rc * 3 * C, cbytes = 2.0 * rc * es + 3.0 * C * C * es;
      const int mt = kb == 2 ? std::min(2, conv_cb_mt((int)l.rows, C, C / 2)) : conv_cb_mt((int)l.rows, C, C);
      const int cwgs = (int)((l.rows + 32 * mt - 1) / (32 * mt)) * (C / 128) * S;
      const CbGnPlan cgp = cb_gn_plan(l.L);
      const int cbdt = u.x3 ? (int)F32X : u.dt;
      ConvCbArgs a;
      a.src_ld = C;
      a.wp = g.conv1.wcb;
      a.slab = p.cbslab;
      a.B = bt;
      a.L = l.L;
      a.C = a.N = C;
      a.kb = kb;
      a.G = G;
      a.eps = 1e-5f;
      a.pf = pf_cb(g.conv2, cwgs);
      if (gnpart_of == cur && wants_gnpart(d)) {
        // the GEMM that produced `cur` left its GroupNorm tile sums: GroupNorm+SiLU rides in conv1's panel prologue, no gn_silu launch
        a.src = cur;
        a.pro = 2;
        a.stats = p.gnpart;
        a.gamma = g.gn1_g;
        a.beta = g.gn1_b;
        timed("conv_cb", cflops + 12.0 * rc, cbytes, [&] { SF_HIP(launch_conv_cb(cbdt, a, s)); });
      } else {
        timed("gn_silu", 12.0 * rc, 3.0 * rc * es,
              [&] { SF_HIP(launch_gn_silu(u.dt, cur, C, bt, l.L, C, G, g.gn1_g, g.gn1_b, 1e-5f, l.act, C, s, pf_cb(g.conv1, bt * G))); });
        a.src = l.act;
        timed("conv_cb", cflops, cbytes, [&] { SF_HIP(launch_conv_cb(cbdt, a, s)); });
      }
      gnpart_of = nullptr;
      timed("cb_reduce_gn", 4.0 * rc, 2.0 * rc * es, [&] {
        SF_HIP(launch_cb_reduce_gn(u.dt, p.cbslab, S, bt, l.L, C, g.conv1.bias, tA, C, G, p.slab, cgp, s, pf_cb(g.conv2, bt * cgp.nch * (C / 128))));
      });
      a.src = tA;
      a.wp = g.conv2.wcb;
      a.pro = 1;
      a.nch = cgp.nch;
      a.chunk_rows = cgp.chunk_rows;
      a.stats = p.slab;
      a.gamma = g.gn2_g;
      a.beta = g.gn2_b;
      a.pf = pf_for(g.inject, cwgs, l.rows);
      timed("conv_cb", cflops + 12.0 * rc, cbytes, [&] { SF_HIP(launch_conv_cb(cbdt, a, s)); });
      timed("cb_reduce_ln", 12.0 * rc, 3.0 * rc * es, [&] {
        SF_HIP(launch_cb_reduce_ln(u.dt, p.cbslab, S, bt, l.L, C, g.conv2.bias, cur, C, p.mod_all + g.mod_off, p.mod_stride, 1e-6f, tA, C, s,
                                   pf_for(g.inject, (int)(l.rows * (C / 4) / 256), l.rows)));
      });
      stats_of = nullptr;
      ConvGemmArgs ai = inject_args(tA, tB);   // InjectChannels on the modulated rows (+ collapsed cross-attention bias when no attention follows)
      if (fuse_attn) {
        ai.rowpart_out = rp_z;
        ai.rowpart_nt = C / 32;
      }
      if (g.attn) ai.pf = pf_for(g.qkv, (int)((l.rows + 31) / 32) * (C / 32), l.rows);
      else if (next && arm_gnpart(g.inject, ai, d)) gnpart_of = tB;   // tB becomes `cur` below
      conv(g.inject, ai, u.dt, u.dt);
    } else {
    conv3(g.conv1, cur, tA, g.gn1_g, g.gn1_b, nullptr, nullptr);
    conv3(g.conv2, tA, tB, g.gn2_g, g.gn2_b, cur, fuse_mod ? rp_y : nullptr, fuse_mod ? &g.inject : nullptr);
    if (fuse_mod) {
      // Modulation + InjectChannels in one GEMM: m = LN_C(y; 1e-6) * (1 + scale) + shift;  z = m + Conv1x1(cat[m, ctx]) (+ bias)
      ConvGemmArgs a = inject_args(tB, tA);
      a.ln_part = rp_y;
      a.ln_nt = C / 32;
      a.ln_ss = p.mod_all + g.mod_off;
      a.ln_ss_ld = p.mod_stride;
      a.ln_eps = 1e-6f;
      a.res_ln = 1;
      if (fuse_attn) {
        a.rowpart_out = rp_z;
        a.rowpart_nt = C / 32;
      }
      if (g.attn) a.pf = pf_for(g.qkv, (int)((l.rows + 31) / 32) * (C / 32), l.rows);
      conv(g.inject, a, u.dt, u.dt, /*ln=*/true);
      std::swap(tA, tB);   // z -> tB, as the code below expects
    } else {
      // Modulation: LN_C(x; eps 1e-6) * (1 + scale) + shift
      timed("ln_modulate", 8.0 * l.rows * C, 2.0 * l.rows * C * dsize(u.dt),
            [&] { SF_HIP(launch_ln_modulate(u.dt, tB, C, p.mod_all + g.mod_off, p.mod_stride, 1e-6f, p.Bt, l.L, C, tA, C, s,
                                            pf_for(g.inject, (int)(l.rows / 4), l.rows))); });
      // InjectChannels: Conv1x1(cat[x, ctx]) + x   (+ collapsed cross-attention bias when no self-attention follows)
      ConvGemmArgs a = inject_args(tA, tB);
      if (fuse_attn) {
        a.rowpart_out = rp_z;
        a.rowpart_nt = C / 32;
      }
      if (g.attn) a.pf = pf_for(g.qkv, (int)((l.rows + 31) / 32) * (C / 32), l.rows);
      conv(g.inject, a, u.dt, u.dt);
    }
    }   // !use_cb
    if (g.attn) {
      // x + W_o MHA(W_q LN_a(x), W_kv LN_b(x)): the two LayerNorms share (mean, rstd); their affines are folded.
      if (fuse_attn) {
        ConvGemmArgs a = qkv_args(tB);
        a.ln_part = rp_z;
        a.ln_nt = C / 32;
        a.ln_eps = 1e-5f;
        a.ln_colsum = g.qkv_colsum;   // raw z through the MFMAs, rstd * (acc - mean * colsum) in the epilogue
        a.pf = pf_for(g.attn_out, (int)((l.rows + 31) / 32) * (3 * u.hd / 32), l.rows);
        conv(g.qkv, a, u.dt, u.dt, /*ln=*/true);
      } else {
        ConvGemmArgs aq = qkv_args(tA);
        // fp32x on the macro tiles: the pre-norm's rows have no other reader and are written pre-split (kernels.h, ConvGemmArgs::src_x3)
        const bool xf = u.x3 && (C % 32) == 0 && conv_gemm_src_x3_ok(filled(g.qkv, aq));
        timed("ln_modulate", 6.0 * l.rows * C, 2.0 * l.rows * C * dsize(u.dt),
              [&] { SF_HIP(launch_ln_modulate(u.dt, tB, C, nullptr, 0, 1e-5f, p.Bt, l.L, C, tA, C, s, Prefetch(), xf)); });
        aq.src_x3 = xf ? 1 : 0;
        conv(g.qkv, aq, u.dt, u.dt);
      }
      attention_tail(g, d, cur, tB, next);
      u.dbg.tap(tapname, u.dt, cur, C, l.rows, C, s);
      return;
    } else {
      void *o = cur;
      cur = tB;
      tB = tA;
      tA = o;
      u.dbg.tap(tapname, u.dt, cur, C, l.rows, C, s);
      return;
    }
    }   // generic (non-thin) resnet / modulation / inject
    if (g.attn) {
      // thin level followed by self-attention: plain pre-norm launch
      timed("ln_modulate", 6.0 * l.rows * C, 2.0 * l.rows * C * dsize(u.dt),
            [&] { SF_HIP(launch_ln_modulate(u.dt, tB, C, nullptr, 0, 1e-5f, p.Bt, l.L, C, tA, C, s)); });
      {
        ConvGemmArgs a;
        a.src = tA;
        a.src_ld = C;
        a.M = (int)l.rows;
        a.Lout = a.Lsrc = l.L;
        a.out = l.qkv;
        a.out_ld = 3 * u.hd;
        conv(g.qkv, a, u.dt, u.dt);
      }
      attention_tail(g, d, cur, tB, next);
      // result in cur's buffer; tA, tB free again
    } else {
      void *o = cur;
      cur = tB;
      tB = tA;
      tA = o;
    }
    u.dbg.tap(tapname, u.dt, cur, C, l.rows, C, s);
  }

  // softmax attention on the packed q | k | v projections, then x' = z + W_o ao (+ collapsed cross-attention bias) -> `out`
  void attention_tail(const Group &g, int d, void *out, const void *z, const Group *next = nullptr) {
    const Level &l = p.lv[d];
    const int C = l.C;
    const size_t es = dsize(u.dt);
    ConvGemmArgs a;
    a.src = l.ao;
    a.src_ld = u.hd;
    a.M = (int)l.rows;
    a.Lout = a.Lsrc = l.L;
    a.out = out;
    a.out_ld = C;
    a.res = z;
    a.res_ld = C;
    if (g.cross) {
      a.badd = p.ca_all + g.ca_off;
      a.badd_ld = u.ca_ld;
    }
    // fp32x on the macro tiles: the attention kernel writes its output rows pre-split for the projection (the only reader)
    const bool xf = u.x3 && (u.hd % 32) == 0 && attention_f32_mfma_ok(3 * u.hd, 3 * u.hd, u.hd, p.Bt, u.cfg.attention_heads) &&
                    conv_gemm_src_x3_ok(filled(g.attn_out, a));
    timed("attention", 4.0 * p.Bt * (double)l.L * l.L * u.hd, 4.0 * l.rows * u.hd * es, [&] {
      SF_HIP(launch_attention(u.dt, l.qkv, 3 * u.hd, static_cast<char *>(l.qkv) + (size_t)u.hd * es, 3 * u.hd, p.Bt, l.L,
                              u.cfg.attention_heads, u.cfg.attention_features, l.ao, u.hd, s, u.x3, xf));
    });
    a.src_x3 = xf ? 1 : 0;
    gnpart_of = nullptr;
    if (next && arm_gnpart(g.attn_out, a, d)) {
      gnpart_of = out;
      if (next->conv1.wcb) a.pf = pf_cb(next->conv1, (int)((l.rows + 31) / 32) * (C / 32));   // the next item starts with conv1 right away
    }
    conv(g.attn_out, a, u.dt, u.dt);
  }

  // Thin level (C = 32 / 64): the item is three conv_thin launches -- conv1 [GN+SiLU in, statistics of h out],
  // conv2 [GN+SiLU in, + x], inject [LayerNorm-modulate in, + itself, (+ cross-attention bias), statistics out] --
  // instead of gn_stats, conv, gn_stats, conv, ln_modulate, conv.  Returns false when the shape is not covered.
  // On return: no attention -> result in cur; attention -> inject output in tB (what the attention code expects).
  bool group_thin(const Group &g, int d, void *&cur, void *&tA, void *&tB) {
    const Level &l = p.lv[d];
    const Block &b = u.blocks[d];
    const int C = l.C, G = u.cfg.resnet_groups;
    if (g.conv1.cin != C || g.inject.cin != C) return false;
    const void *w1 = g.conv1.direct ? g.conv1.wt : g.conv1.w, *w2 = g.conv2.direct ? g.conv2.wt : g.conv2.w;
    const void *w3 = g.inject.direct ? g.inject.wt : g.inject.w;
    if (!w1 || !w2 || !w3) return false;
    const ThinPlan tp = conv_thin_plan(p.Bt, l.L, C);
    ConvThinArgs base;
    base.B = p.Bt;
    base.L = base.Ls = l.L;
    base.C = base.N = C;
    base.G = G;
    base.rw = tp.rw;
    base.nchw = tp.nchw;
    base.nch_in = tp.nchw;
    base.chunk_in = tp.rw;
    base.src_ld = base.out_ld = base.res_ld = C;
    base.x3 = u.x3 ? 1 : 0;
    ConvThinArgs a1 = base, a3 = base;
    a1.taps = 3;
    a1.pro = 1;
    a3.taps = 1;
    a3.pro = 2;
    a3.C2 = g.inject.direct ? pad_to(g.inject.cin2, 8) : g.inject.cin2;
    if (a3.C2 > b.ctx_ld) return false;
    if (!conv_thin_supported(u.dt, a1) || !conv_thin_supported(u.dt, a3)) return false;
    float *sA = p.slab, *sB = p.slab + p.slab_half;
    const double es = dsize(u.dt), rc = (double)l.rows * C;
    if (stats_of != cur)
      timed("gn_stats", 3.0 * rc, rc * es, [&] { SF_HIP(launch_gn_stats(u.dt, cur, C, p.Bt, l.L, C, G, tp.nchw, tp.rw, sA, s)); });
    {
      ConvThinArgs a = a1;
      a.src = cur;
      a.w = w1;
      if (g.conv1.direct) a.w32 = static_cast<const float *>(g.conv1.w);
      a.bias = g.conv1.bias;
      a.gamma = g.gn1_g;
      a.beta = g.gn1_b;
      a.stats_in = sA;
      a.stats_out = sB;
      a.out = tA;
      timed("conv_thin", 2.0 * rc * 3 * C, 2.0 * rc * es + 3.0 * C * C * es, [&] { SF_HIP(launch_conv_thin(u.dt, a, s)); });
    }
    // conv2 + Modulation + InjectChannels in one launch when the fused tail covers the shape
    {
      ThinTailArgs t;
      t.x3 = u.x3 ? 1 : 0;
      t.h = tA;
      t.x = cur;
      t.ctx = l.ctx;
      t.ctx_ld = b.ctx_ld;
      t.w2 = w2;
      if (g.conv2.direct && g.inject.direct) {
        t.w2_32 = static_cast<const float *>(g.conv2.w);
        t.w3_32 = static_cast<const float *>(g.inject.w);
        t.c2real = g.inject.cin2;
      }
      t.bias2 = g.conv2.bias;
      t.gamma = g.gn2_g;
      t.beta = g.gn2_b;
      t.stats_in = sB;
      t.w3 = w3;
      t.bias3 = g.inject.bias;
      t.ss = p.mod_all + g.mod_off;
      t.ss_ld = p.mod_stride;
      if (g.cross && !g.attn) {
        t.badd = p.ca_all + g.ca_off;
        t.badd_ld = u.ca_ld;
      }
      t.out = tB;
      t.stats_out = g.attn ? nullptr : sA;
      t.B = p.Bt;
      t.L = l.L;
      t.C = C;
      t.C2 = a3.C2;
      t.G = G;
      t.nch_in = tp.nchw;
      t.chunk_in = tp.rw;
      t.rw = tp.rw;
      t.nchw = tp.nchw;
      if (!u.no_thin_tail && g.conv2.bias && g.inject.bias && thin_tail_supported(u.dt, t)) {
        const double kin = g.inject.kreal > 0 ? g.inject.kreal : g.inject.K;
        timed("conv_thin", 2.0 * rc * 3 * C + 2.0 * rc * kin + 8.0 * rc, 3.0 * rc * es + (double)l.rows * (kin - C) * es + (3.0 * C + kin) * C * es,
              [&] { SF_HIP(launch_thin_tail(u.dt, t, s)); });
        if (g.attn) {
          stats_of = nullptr;   // z in tB, tA free: what the attention code expects
        } else {
          void *o = cur;
          cur = tB;
          tB = o;
          stats_of = cur;
        }
        return true;
      }
    }
    {
      ConvThinArgs a = a1;
      a.src = tA;
      a.w = w2;
      if (g.conv2.direct) a.w32 = static_cast<const float *>(g.conv2.w);
      a.bias = g.conv2.bias;
      a.gamma = g.gn2_g;
      a.beta = g.gn2_b;
      a.stats_in = sB;
      a.res = cur;
      a.out = tB;
      timed("conv_thin", 2.0 * rc * 3 * C, 3.0 * rc * es + 3.0 * C * C * es, [&] { SF_HIP(launch_conv_thin(u.dt, a, s)); });
    }
    {
      ConvThinArgs a = a3;
      a.src = tB;
      a.src2 = l.ctx;
      a.src2_ld = b.ctx_ld;
      a.w = w3;
      a.bias = g.inject.bias;
      a.ss = p.mod_all + g.mod_off;
      a.ss_ld = p.mod_stride;
      a.eps = 1e-6f;
      a.res_self = 1;
      if (g.cross && !g.attn) {
        a.badd = p.ca_all + g.ca_off;
        a.badd_ld = u.ca_ld;
      }
      a.stats_out = g.attn ? nullptr : sA;
      a.out = tA;
      const double kin = g.inject.kreal > 0 ? g.inject.kreal : g.inject.K;
      timed("conv_thin", 2.0 * rc * kin + 8.0 * rc, 2.0 * rc * es + (double)l.rows * (kin - C) * es + kin * C * es,
            [&] { SF_HIP(launch_conv_thin(u.dt, a, s)); });
    }
    if (g.attn) {
      std::swap(tA, tB);   // inject output -> tB, tA free
      stats_of = nullptr;
    } else {
      void *o = cur;
      cur = tA;
      tA = o;
      stats_of = cur;
    }
    return true;
  }

  // Block d: skip + scale * Up(items_up(inner(items_down(Down(x)))))
  // Returns true when the GroupNorm partials of xout were left in p.slab (thin up convolution).
  bool block(int d, const void *xin, int xin_dt, void *xout, int xout_dt) {
    const sf_unet_config &c = u.cfg;
    const Block &b = u.blocks[d];
    Level &l = p.lv[d];
    const int Lprev = l.L * b.factor;
    void *cur = l.buf[0], *tA = l.buf[1], *tB = l.buf[2];
    stats_of = nullptr;
    cur_depth = d;
    bool down_done = false;
    if (!b.down.direct && xin_dt == u.dt) {   // patchify conv on the (rows/f, f*cin) view as a thin-level kernel
      const ThinPlan tp = conv_thin_plan(p.Bt, l.L, l.C);
      ConvThinArgs a;
      a.x3 = u.x3 ? 1 : 0;
      a.B = p.Bt;
      a.L = a.Ls = l.L;
      a.C = b.factor * b.cin;
      a.N = l.C;
      a.taps = 1;
      a.G = c.resnet_groups;
      a.rw = tp.rw;
      a.nchw = tp.nchw;
      a.src = xin;
      a.src_ld = a.C;
      a.w = b.down.w;
      a.bias = b.down.bias;
      a.out = cur;
      a.out_ld = l.C;
      a.stats_out = p.slab;
      if (!conv_thin_supported(u.dt, a)) a.stats_out = nullptr;
      if (conv_thin_supported(u.dt, a)) {
        const double es = dsize(u.dt), kin = b.down.kreal > 0 ? b.down.kreal : b.down.K;
        timed("conv_thin", 2.0 * l.rows * l.C * kin, (double)l.rows * (kin + l.C) * es + kin * l.C * es,
              [&] { SF_HIP(launch_conv_thin(u.dt, a, s)); });
        if (a.stats_out) stats_of = cur;
        down_done = true;
      }
    }
    if (!down_done) {
      ConvGemmArgs a;
      a.src = xin;
      a.M = (int)l.rows;
      a.out = cur;
      a.out_ld = l.C;
      if (b.down.direct) {
        a.src_ld = b.cin;
        a.Lout = l.L;
        a.Lsrc = Lprev;
        a.stride = b.factor;
        a.pad = 0;
      } else {
        a.src_ld = b.factor * b.cin;  // (rows/f, f*cin) view: taps folded into channels
        a.Lout = a.Lsrc = l.L;
      }
      ConvW w = b.down;
      if (!w.direct) {
        w.cin = b.factor * b.cin;
        w.taps = 1;
      }
      gnpart_of = nullptr;
      if (!w.direct && !b.down_items.empty() && arm_gnpart(w, a, d)) gnpart_of = cur;
      conv(w, a, xin_dt, u.dt);
    }
    const std::string pre = "d" + std::to_string(d);
    u.dbg.tap(pre + ".down", u.dt, cur, l.C, l.rows, l.C, s);
    const bool innermost = d + 1 >= c.n_layers;
    for (size_t j = 0; j < b.down_items.size(); ++j) {
      // the consumer of this item's output: the next down item; at the innermost level the first up item; else the next level's down conv
      const Group *nx = j + 1 < b.down_items.size() ? &b.down_items[j + 1] : ((innermost && !b.up_items.empty()) ? &b.up_items[0] : nullptr);
      group(b.down_items[j], d, cur, tA, tB, pre + ".items_down." + std::to_string(j), nx);
    }
    if (d + 1 < c.n_layers) {
      up_gnpart = false;
      const bool have = block(d + 1, cur, u.dt, tA, u.dt);
      cur_depth = d;
      void *o = cur;
      cur = tA;
      tA = o;
      stats_of = have ? cur : nullptr;   // deeper levels share the statistics slabs
      gnpart_of = (up_gnpart && !b.up_items.empty()) ? cur : nullptr;
      up_gnpart = false;
    }   // (innermost level: the last down item feeds the first up item directly; its tile sums are still valid)
    for (size_t j = 0; j < b.up_items.size(); ++j)
      group(b.up_items[j], d, cur, tA, tB, pre + ".items_up." + std::to_string(j), j + 1 < b.up_items.size() ? &b.up_items[j + 1] : nullptr);
    bool up_stats = false, up_done = false;
    if (b.up_transposed) {   // x_out = skip + scale * ConvTranspose(h): one GEMM onto the (rows, f * cin) view of the outer level
      ConvGemmArgs a;
      a.src = cur;
      a.src_ld = l.C;
      a.M = (int)l.rows;
      a.Lout = a.Lsrc = l.L;
      a.out = xout;
      a.out_ld = b.factor * b.cin;
      a.res = xin;
      a.res_ld = b.factor * b.cin;
      a.bscale = p.mod_all + b.skip_off;
      a.bscale_ld = p.mod_stride;
      if (xout_dt != u.dt && !b.up.direct) fail(SF_ERR_UNSUPPORTED, "depth 0 must be a thin level (channels[0] %% 32 != 0)");
      conv(b.up, a, u.dt, xout_dt);
      up_done = true;
    }
    if (!up_done && !b.up.direct && xout_dt == u.dt && xin_dt == u.dt) {   // nearest-upsample + conv3 + SkipModulate as a thin-level kernel
      const ThinPlan tp = conv_thin_plan(p.Bt, Lprev, b.cin);
      ConvThinArgs a;
      a.x3 = u.x3 ? 1 : 0;
      a.B = p.Bt;
      a.L = Lprev;
      a.Ls = l.L;
      a.up_shift = b.up_shift;
      a.C = l.C;
      a.N = b.cin;
      a.taps = 3;
      a.G = c.resnet_groups;
      a.rw = tp.rw;
      a.nchw = tp.nchw;
      a.src = cur;
      a.src_ld = l.C;
      a.w = b.up.w;
      a.bias = b.up.bias;
      a.bscale = p.mod_all + b.skip_off;
      a.bscale_ld = p.mod_stride;
      a.res = xin;
      a.res_ld = b.cin;
      a.out = xout;
      a.out_ld = b.cin;
      a.stats_out = p.slab;
      if (!conv_thin_supported(u.dt, a)) a.stats_out = nullptr;
      if ((1 << b.up_shift) == b.factor && conv_thin_supported(u.dt, a)) {
        const double es = dsize(u.dt), rows_out = (double)l.rows * b.factor;
        timed("conv_thin", 2.0 * rows_out * b.cin * 3 * l.C, ((double)l.rows * l.C + 2.0 * rows_out * b.cin) * es + 3.0 * l.C * b.cin * es,
              [&] { SF_HIP(launch_conv_thin(u.dt, a, s)); });
        up_stats = a.stats_out != nullptr;
        up_done = true;
      }
    }
    if (!up_done) {
      ConvGemmArgs a;
      a.src = cur;
      a.src_ld = l.C;
      a.M = (int)(l.rows * b.factor);
      a.Lout = Lprev;
      a.Lsrc = l.L;
      a.stride = 1;
      a.pad = 1;
      a.up_shift = b.up_shift;
      a.out = xout;
      a.out_ld = b.cin;
      a.res = xin;
      a.res_ld = b.cin;
      a.bscale = p.mod_all + b.skip_off;
      a.bscale_ld = p.mod_stride;
      if (xout_dt != u.dt && !b.up.direct) fail(SF_ERR_UNSUPPORTED, "depth 0 must be a thin level (channels[0] %% 32 != 0)");
      if (d > 0 && xout_dt == u.dt && !b.up.direct && arm_gnpart(b.up, a, d - 1)) up_gnpart = true;
      conv(b.up, a, u.dt, xout_dt);
    }
    u.dbg.tap(pre + ".out", xout_dt, xout, b.cin, l.rows * b.factor, b.cin, s);
    return up_stats;
  }

  // features + modulation vectors of one step; sigma from sig[b] (sig_idx == nullptr) or sig[*sig_idx]
  void features(const float *sig, const int *sig_idx) { features_rows(sig, sig_idx, p.Bt, p.mod_all); }
  // rows x (time MLP -> all Modulation / SkipModulate vectors); row r uses sigma sig[r] (or sig[*sig_idx])
  void features_rows(const float *sig, const int *sig_idx, int rows, float *mod_out) {
    timed("time_fourier", 0.0, (double)rows * u.four_ld * dsize(u.dt),
          [&] { SF_HIP(launch_time_fourier(u.dt, sig, sig_idx, u.fourier_w, rows, u.half, p.four, u.four_ld, s)); });
    dense(u.lin0, p.four, u.four_ld, rows, p.f1, u.mf, /*gelu unless the [RECALLED] switch says none*/ u.cfg.time_no_first_act ? 0 : 2, false);
    dense(u.mlp0, p.f1, u.mf, rows, p.f2, u.mf, 2, false);
    dense(u.mlp1, p.f2, u.mf, rows, p.sf, u.mf, /*silu(gelu)*/ 3, false);
    dense(u.mod, p.sf, u.mf, rows, mod_out, u.mod_ld, 0, true);
  }

  // per-call conditioning: context pyramids to channels-last, cross-attention collapse
  void conditioning(const float *const *ctx, const float *emb) {
    const sf_unet_config &c = u.cfg;
    const size_t es = dsize(u.dt);
    for (int d = 0; d < c.n_layers; ++d) {
      const Level &l = p.lv[d];
      const Block &b = u.blocks[d];
      SF_HIP(launch_cf_to_cl(u.dt, ctx[d], p.B, b.ctx, l.L, l.ctx, b.ctx_ld, s));
      if (p.two)
        SF_HIP(hipMemcpyAsync(static_cast<char *>(l.ctx) + (size_t)p.B * l.L * b.ctx_ld * es, l.ctx, (size_t)p.B * l.L * b.ctx_ld * es,
                              hipMemcpyDeviceToDevice, s));
    }
    if (u.n_ca == 0) return;
    const int E = c.embedding_features;
    SF_HIP(hipMemcpyAsync(p.emb2, emb, (size_t)p.B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (p.two)  // unconditional rows use the learned FixedEmbedding (ClassifierFreeGuidancePlugin)
      for (int b = 0; b < p.B; ++b)
        SF_HIP(hipMemcpyAsync(p.emb2 + (size_t)(p.B + b) * E, u.fixed_emb, E * sizeof(float), hipMemcpyDeviceToDevice, s));
    // LN(e) without affine (folded into wv_cat); ln_modulate wants the compute type
    SF_HIP(launch_pack_rows(u.dt, p.emb2, p.Bt, E, E, nullptr, p.emb_t, E, s));
    SF_HIP(launch_ln_modulate(u.dt, p.emb_t, E, nullptr, 0, 1e-5f, p.Bt, 1, E, p.xhat_e, E, s));
    dense(u.wv_cat, p.xhat_e, E, p.Bt, p.v_all, u.n_ca * u.hd, 0, false);
    if (u.ca_nblocks > 0) {   // all per-item output projections in one launch
      SF_HIP(launch_cross_out_grouped(u.dt, u.ca_items, u.ca_blocks, u.ca_nblocks, p.v_all, u.n_ca * u.hd, p.Bt, u.hd, p.ca_all, u.ca_ld, s));
      ++u.launches;
      return;
    }
    for (int d = 0; d < c.n_layers; ++d) {
      const Block &b = u.blocks[d];
      for (int pass = 0; pass < 2; ++pass)
        for (const Group &g : (pass == 0 ? b.down_items : b.up_items)) {
          if (!g.cross) continue;
          dense(g.cross_out, static_cast<char *>(p.v_all) + (size_t)g.ca_idx * u.hd * es, u.n_ca * u.hd, p.Bt,
                p.ca_all + g.ca_off, u.ca_ld, 0, true);
        }
    }
  }

  // the rows [br*Bt/nbr, (br+1)*Bt/nbr) of every per-clip buffer
  Plan branch_view(int br) const {
    Plan v = p;
    const size_t es = dsize(u.dt);
    const int bt = p.Bt / p.nbr;
    v.Bt = bt;
    v.nbr = 1;
    for (size_t d = 0; d < v.lv.size(); ++d) {
      Level &l = v.lv[d];
      l.rows = (int64_t)bt * l.L;
      const int64_t r0 = (int64_t)br * l.rows;
      auto off = [&](void *ptr, int64_t cols) { return ptr ? static_cast<void *>(static_cast<char *>(ptr) + r0 * cols * es) : nullptr; };
      for (int i = 0; i < 3; ++i) l.buf[i] = off(l.buf[i], l.C);
      l.act = off(l.act, l.C);
      l.qkv = off(l.qkv, 3 * u.hd);
      l.ao = off(l.ao, u.hd);
      l.ctx = off(l.ctx, u.blocks[d].ctx_ld);
    }
    const int64_t n0 = (int64_t)br * bt * p.L0 * u.cfg.in_channels;
    v.x2 = p.x2 + n0;
    v.vout = p.vout + n0;
    v.mod_all = p.mod_all + (int64_t)br * bt * p.mod_stride;
    v.ca_all = p.ca_all + (int64_t)br * bt * u.ca_ld;
    v.slab = p.slab + (int64_t)br * p.slab_stride;
    v.rowpart = p.rowpart + (int64_t)br * p.rowpart_stride;
    v.cbslab = p.cbslab ? p.cbslab + (int64_t)br * p.cbslab_stride : nullptr;
    v.gnpart = p.gnpart ? p.gnpart + (int64_t)br * p.gnpart_stride : nullptr;
    return v;
  }

  // one U-Net evaluation of the (possibly doubled) batch: x (B rows) -> p.vout (Bt rows)
  void eval(const float *x, const float *sig, const int *sig_idx, bool features_ready = false) {
    const int64_t n = (int64_t)p.B * p.L0 * u.cfg.in_channels;
    // classifier-free guidance evaluates [x ; x] as one 2B batch; a single pass reads the caller's x in place
    float *xin = const_cast<float *>(x);
    if (p.two) {
      SF_HIP(hipMemcpyAsync(p.x2, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
      SF_HIP(hipMemcpyAsync(p.x2 + n, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
      xin = p.x2;
    }
    if (!features_ready) features(sig, sig_idx);
    if (p.nbr <= 1) {
      block(0, xin, F32, p.vout, F32);
      return;
    }
    const bool serial = u.prof_on;   // instrumented pass: same kernel shapes, one after another on the launch stream
    if (!serial) {
      u.ensure_branch_streams(p.nbr);
      SF_HIP(hipEventRecord(u.ev_fork, s));
    }
    for (int br = 0; br < p.nbr; ++br) {
      Plan v = branch_view(br);
      hipStream_t sb = (serial || br == 0) ? s : u.bstream[br];
      if (sb != s) SF_HIP(hipStreamWaitEvent(sb, u.ev_fork, 0));
      Exec eb{u, v, sb};
      eb.block(0, xin + (v.x2 - p.x2), F32, v.vout, F32);
      if (sb != s) SF_HIP(hipEventRecord(u.ev_join[br], sb));
    }
    if (!serial)
      for (int br = 1; br < p.nbr; ++br) SF_HIP(hipStreamWaitEvent(s, u.ev_join[br], 0));
  }
};

void check_ws(const sf_unet *h, void *ws, int64_t ws_bytes, int B, int L0, bool two, int steps) {
  if (!ws) fail(SF_ERR_WORKSPACE, "workspace is null");
  Workspace dry(nullptr, 0);
  make_plan(*h, dry, B, L0, two, steps);
  if (dry.used() > ws_bytes) fail(SF_ERR_WORKSPACE, "workspace too small: need %lld bytes, have %lld", (long long)dry.used(), (long long)ws_bytes);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
#define SF_API_BEGIN try {
#define SF_API_END                 \
  }                                \
  catch (const EngineError &e) {   \
    return e.code;                 \
  }                                \
  catch (const std::exception &e) {\
    set_error("%s", e.what());     \
    return SF_ERR_INVALID;         \
  }

extern "C" {

int sf_unet_create(const sf_unet_config *cfg, const sf_tensor *weights, int n_weights, void *stream, sf_unet **out) {
  SF_API_BEGIN
  if (!cfg || !out || (!weights && n_weights > 0)) fail(SF_ERR_INVALID, "null argument");
  *out = nullptr;
  std::unique_ptr<sf_unet> u(new sf_unet());
  u->cfg = *cfg;
  WeightMap wm(weights, n_weights);
  build(*u, &wm, static_cast<hipStream_t>(stream));
  *out = u.release();
  return SF_OK;
  SF_API_END
}

void sf_unet_destroy(sf_unet *h) { delete h; }

static int list_params(const sf_unet_config *cfg, std::vector<std::pair<std::string, int64_t>> &names) {
  SF_API_BEGIN
  if (!cfg) fail(SF_ERR_INVALID, "null config");
  sf_unet u;
  u.cfg = *cfg;
  u.listing = true;
  build(u, nullptr, nullptr);
  names = u.names;
  return SF_OK;
  SF_API_END
}

int sf_unet_param_count(const sf_unet_config *cfg) {
  std::vector<std::pair<std::string, int64_t>> names;
  if (list_params(cfg, names) != SF_OK) return -1;
  return (int)names.size();
}

int sf_unet_param_name(const sf_unet_config *cfg, int index, char *name_out, int name_cap, int64_t *numel_out) {
  std::vector<std::pair<std::string, int64_t>> names;
  int rc = list_params(cfg, names);
  if (rc != SF_OK) return rc;
  if (index < 0 || index >= (int)names.size() || !name_out || name_cap <= 0) return SF_ERR_INVALID;
  snprintf(name_out, name_cap, "%s", names[index].first.c_str());
  if (numel_out) *numel_out = names[index].second;
  return SF_OK;
}

int64_t sf_unet_workspace_bytes(const sf_unet *h, int B, int L0, int two_pass) {
  try {
    if (!h) fail(SF_ERR_INVALID, "null handle");
    Workspace dry(nullptr, 0);
    make_plan(*h, dry, B, L0, two_pass != 0, 1);
    return dry.used();
  } catch (const EngineError &) {
    return -1;
  }
}

int64_t sf_vsample_workspace_bytes(const sf_unet *h, int B, int L0, int two_pass, int num_steps) {
  try {
    if (!h) fail(SF_ERR_INVALID, "null handle");
    if (num_steps < 1) fail(SF_ERR_INVALID, "num_steps must be >= 1");
    Workspace dry(nullptr, 0);
    make_plan(*h, dry, B, L0, two_pass != 0, num_steps);
    return dry.used();
  } catch (const EngineError &) {
    return -1;
  }
}

int sf_unet_forward(sf_unet *h, const float *x, const float *sigma, const float *const *ctx, const float *emb, int B, int L0,
                    float embedding_scale, float *out, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!h || !x || !sigma || !ctx || !out) fail(SF_ERR_INVALID, "null argument");
  if (!emb) fail(SF_ERR_INVALID, "ClassifierFreeGuidancePlugin requires embedding");
  const bool two = embedding_scale != 1.0f;
  check_ws(h, ws, ws_bytes, B, L0, two, 1);
  Workspace w(ws, ws_bytes);
  Plan p = make_plan(*h, w, B, L0, two, 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  Exec ex{*h, p, s};
  h->dbg.reset();
  ex.conditioning(ctx, emb);
  h->launches = 0;
  // sigma for the doubled batch: rows [B, 2B) repeat rows [0, B)
  float *sig2 = p.sigs + 1;
  SF_HIP(hipMemcpyAsync(sig2, sigma, B * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (two) SF_HIP(hipMemcpyAsync(sig2 + B, sigma, B * sizeof(float), hipMemcpyDeviceToDevice, s));
  h->prof.clear();
  h->ev_used = 0;
  if (h->prof_on) {
    // calibration record: the same event pair around an (almost) empty kernel = the fixed cost the HIP-event method
    // adds to every launch (event processing + dispatch latency); bench.py subtracts it
    int *scratch = p.step + 1;
    for (int i = 0; i < 4; ++i) ex.timed("calib_empty", 0.0, 0.0, [&] { SF_HIP(launch_step_advance(scratch, s)); });
  }
  ex.eval(x, sig2, nullptr);
  if (h->prof_on) {
    SF_HIP(hipStreamSynchronize(s));
    for (auto &r : h->prof) SF_HIP(hipEventElapsedTime(&r.ms, r.e0, r.e1));
  }
  const int64_t n = (int64_t)B * L0 * h->cfg.in_channels;
  if (two) SF_HIP(launch_cfg_combine(p.vout, p.vout + n, embedding_scale, out, n, s));
  else SF_HIP(hipMemcpyAsync(out, p.vout, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return SF_OK;
  SF_API_END
}

int sf_vsample(sf_unet *h, float *x, const float *const *ctx, const float *emb, int B, int L0, int num_steps, float embedding_scale,
               int use_graph, void *ws, int64_t ws_bytes, void *stream) {
  SF_API_BEGIN
  if (!h || !x || !ctx) fail(SF_ERR_INVALID, "null argument");
  if (!emb) fail(SF_ERR_INVALID, "ClassifierFreeGuidancePlugin requires embedding");
  if (num_steps < 1) fail(SF_ERR_INVALID, "num_steps must be >= 1");
  const bool two = embedding_scale != 1.0f;
  check_ws(h, ws, ws_bytes, B, L0, two, num_steps);
  Workspace w(ws, ws_bytes);
  Plan p = make_plan(*h, w, B, L0, two, num_steps);
  hipStream_t user = static_cast<hipStream_t>(stream);
  hipStream_t s = user;
  if (use_graph && num_steps > 1) {
    if (!h->own_stream) {
      SF_HIP(sf_unet::make_branch_stream(&h->own_stream, 0, p.nbr));
      SF_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
      SF_HIP(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    }
    SF_HIP(hipEventRecord(h->ev_in, user));
    SF_HIP(hipStreamWaitEvent(h->own_stream, h->ev_in, 0));
    s = h->own_stream;
  }
  Exec ex{*h, p, s};
  h->dbg.reset();
  // taps are a forward()-only facility: cleared for the call, restored on EVERY exit path
  struct DbgGuard {
    DebugTaps &d;
    float *saved;
    ~DbgGuard() { d.buf = saved; }
  } dbg_guard{h->dbg, h->dbg.buf};
  h->dbg.buf = nullptr;
  // A failure below (EngineError, HIP error, failed capture) must not leave work queued on the engine's streams, nor a
  // half-valid graph cache: drain the streams, drop the cached graphs, then report the error.
  struct FailGuard {
    sf_unet *h;
    bool armed = true;
    ~FailGuard() {
      if (!armed) return;
      if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
      for (hipStream_t b : h->bstream)
        if (b) (void)hipStreamSynchronize(b);
      h->drop_all_graphs();
      (void)hipGetLastError();
    }
  } fail_guard{h};
  static const bool timing = getenv("SF_TIMING") != nullptr;   // debugging aid: host-side phase times on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
  const auto t_begin = now();
  ex.conditioning(ctx, emb);
  if (timing) {
    SF_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "[sf_vsample] conditioning %.3f ms\n", ms_since(t_begin));
  }

  // LinearSchedule(1 -> 0) and (alpha, beta) = (cos, sin)(sigma*pi/2), exactly as VSampler builds them in fp32
  // (torch.linspace: start + i*step for the first half, end - (steps-1-i)*step for the second).
  const int T = num_steps;
  std::vector<float> sig(T + 1), host(4 * (size_t)T + (size_t)T);
  {
    const float step = (0.0f - 1.0f) / (float)T;
    const int halfway = (T + 1) / 2;
    for (int i = 0; i <= T; ++i) sig[i] = (i < halfway) ? (1.0f + step * (float)i) : (0.0f - step * (float)(T - i));
  }
  const float hp = (float)(M_PI / 2.0);
  for (int i = 0; i < T; ++i) {
    float a0 = cosf(sig[i] * hp), b0 = sinf(sig[i] * hp), a1 = cosf(sig[i + 1] * hp), b1 = sinf(sig[i + 1] * hp);
    host[4 * i + 0] = a0;
    host[4 * i + 1] = b0;
    host[4 * i + 2] = a1;
    host[4 * i + 3] = b1;
    host[4 * (size_t)T + i] = sig[i];
  }
  float *sched = p.sched;
  float *sigs = p.sigs;
  SF_HIP(hipMemcpyAsync(sched, host.data(), 4 * (size_t)T * sizeof(float), hipMemcpyHostToDevice, s));
  SF_HIP(hipMemcpyAsync(sigs, host.data() + 4 * (size_t)T, (size_t)T * sizeof(float), hipMemcpyHostToDevice, s));
  SF_HIP(hipMemsetAsync(p.step, 0, 16 * sizeof(int), s));

  const int64_t n = (int64_t)B * L0 * h->cfg.in_channels;
  // sigma_i is the same for every clip, so the time MLP and the 42 Modulation / SkipModulate projections of ALL steps
  // are one set of GEMMs with M = num_steps, once per call (instead of five launches inside every step); a step then
  // starts by selecting its row (which also advances the device step counter) and every consumer reads that one row
  // for all clips (clip stride 0).
  const bool pre = T > 1 && p.mod_steps != nullptr;
  if (pre) {
    ex.features_rows(sigs, nullptr, T, p.mod_steps);
    p.mod_stride = 0;
  }
  float *xs = p.xs;
  SF_HIP(hipMemcpyAsync(xs, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  auto one_step = [&]() {
    if (pre) {
      SF_HIP(launch_step_select(p.mod_steps, h->mod_ld, p.step, p.mod_all, s));
      ex.eval(xs, nullptr, nullptr, /*features_ready=*/true);
      SF_HIP(launch_vsampler_update(xs, p.vout, two ? p.vout + n : nullptr, embedding_scale, sched - 4, p.step, n, s));   // *step == i + 1 here
    } else {
      ex.eval(xs, sigs, p.step);
      SF_HIP(launch_vsampler_update(xs, p.vout, two ? p.vout + n : nullptr, embedding_scale, sched, p.step, n, s));
      SF_HIP(launch_step_advance(p.step, s));
    }
  };

  sf_unet::GraphKey key;
  key.B = B;
  key.L0 = L0;
  key.T = T > kSchedSteps ? T : 0;   // the layout (hence the graph) depends on num_steps only beyond the fixed schedule capacity
  key.nbr = p.nbr;
  key.two = two;
  key.scale = embedding_scale;
  key.ws = ws;
  key.valid = true;
  if (use_graph && T > 1 && !h->prof_on) h->activate_graphs(key);
  // One host synchronisation per call, placed HERE: the pageable H2D copies of the schedule above must have been staged
  // before `host` dies, and the step graphs are measurably faster when their streams are idle at the first launch
  // (449 vs 408 steps/s at 50 steps: launching onto busy streams leaves the two branch pipelines out of phase).
  SF_HIP(hipStreamSynchronize(s));
  if (timing) fprintf(stderr, "[sf_vsample] + schedule, per-step features: %.3f ms since entry\n", ms_since(t_begin));
  // Independent branch pipelines.  Without guidance the clip slices never interact during the whole loop, so each
  // branch owns a step counter, a modulation row and a SINGLE-STREAM step graph replayed on its own stream; the streams
  // meet only at the end of the call.  (A two-stream graph with a fork/join per step costs ~1.7 ms of host time per launch
  // on this runtime -- node by node -- against ~0.07 ms for a single-stream graph, and joins the branches 50 times.)
  const bool indep = use_graph && pre && !two && p.nbr > 1 && !h->prof_on && !h->no_indep_branches;
  if (indep) {
    h->ensure_branch_streams(p.nbr);
    const int64_t nb = n / p.nbr;
    auto stream_of = [&](int br) { return br == 0 ? s : h->bstream[br]; };
    auto branch_step = [&](int br) {
      hipStream_t sb = stream_of(br);
      Plan v = ex.branch_view(br);
      v.mod_all = p.mod_all + (int64_t)br * h->mod_ld;   // the branch's own copy of the step's modulation row
      SF_HIP(launch_step_select(p.mod_steps, h->mod_ld, p.step + br, v.mod_all, sb));
      Exec eb{*h, v, sb};
      float *xb = xs + (v.x2 - p.x2);
      eb.block(0, xb, F32, v.vout, F32);
      SF_HIP(launch_vsampler_update(xb, v.vout, nullptr, embedding_scale, sched - 4, p.step + br, nb, sb));
    };
    SF_HIP(hipEventRecord(h->ev_fork, s));
    for (int br = 1; br < p.nbr; ++br) SF_HIP(hipStreamWaitEvent(h->bstream[br], h->ev_fork, 0));
    const bool cached = h->gexec_indep && h->gkey == key;
    const auto t_l = now();
    if (!cached) {
      h->gkey.valid = false;
      h->launches = 0;
      for (int br = 0; br < p.nbr; ++br) branch_step(br);   // step 0 eagerly (one-time kernel attribute setup outside capture)
      for (int br = 0; br < p.nbr; ++br) {
        hipStream_t sb = stream_of(br);
        hipGraph_t graph = nullptr;
        SF_HIP(hipStreamBeginCapture(sb, hipStreamCaptureModeThreadLocal));
        try {
          branch_step(br);
        } catch (...) {
          hipGraph_t g2 = nullptr;
          (void)hipStreamEndCapture(sb, &g2);
          if (g2) (void)hipGraphDestroy(g2);
          throw;
        }
        SF_HIP(hipStreamEndCapture(sb, &graph));
        if (h->gexec_br[br]) {
          (void)hipGraphExecDestroy(h->gexec_br[br]);
          h->gexec_br[br] = nullptr;
        }
        ++h->graph_captures;
        hipError_t e = hipGraphInstantiate(&h->gexec_br[br], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) fail(SF_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      }
      h->gexec_indep = true;
      h->gkey = key;
    }
    {
      static const double stagger_us = [] {   // tuning aid: start branch b  b * stagger microseconds late
        const char *e = tune_env("SF_BRANCH_STAGGER_US");
        return e ? atof(e) : 0.0;
      }();
      if (stagger_us > 0.0)
        for (int br = 1; br < p.nbr; ++br) SF_HIP(launch_spin(stagger_us * br, stream_of(br)));
    }
    // (the capture above only records step 1; replay it for the remaining steps -- all of them when the graphs were cached)
    for (int i = cached ? 0 : 1; i < T; ++i)
      for (int br = 0; br < p.nbr; ++br) SF_HIP(hipGraphLaunch(h->gexec_br[br], stream_of(br)));
    for (int br = 1; br < p.nbr; ++br) {
      SF_HIP(hipEventRecord(h->ev_join[br], h->bstream[br]));
      SF_HIP(hipStreamWaitEvent(s, h->ev_join[br], 0));
    }
    const double enq = ms_since(t_l);
    // The call returns when the loop has finished.  Measured: with ~100 graph launches still queued on two streams, a
    // caller that goes on to hipDeviceSynchronize (torch.cuda.synchronize) slows the GPU side by ~10 % (404-418 vs 448
    // steps/s); waiting on the one stream that joins the branches does not.
    SF_HIP(hipStreamSynchronize(s));
    if (timing) fprintf(stderr, "[sf_vsample] %d steps x %d independent branch graphs: enqueue %.3f ms, done after %.3f ms\n", T, p.nbr, enq, ms_since(t_l));
  } else if (use_graph && T > 1 && h->gexec && !h->gexec_indep && !h->prof_on && h->gkey == key) {
    const auto t_l = now();
    for (int i = 0; i < T; ++i) SF_HIP(hipGraphLaunch(h->gexec, s));   // steady state: no eager step, no capture
    if (timing) {
      const double enq = ms_since(t_l);
      SF_HIP(hipStreamSynchronize(s));
      fprintf(stderr, "[sf_vsample] %d graph launches: enqueue %.3f ms, done after %.3f ms\n", T, enq, ms_since(t_l));
    }
  } else {
  h->launches = 0;
  one_step();  // step 0 eagerly (also performs every one-time kernel attribute setup outside capture)
  h->launches += 2;
  if (T > 1) {
    if (use_graph) {
      h->gkey.valid = false;
      h->gexec_indep = false;
      hipGraph_t graph = nullptr;
      SF_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      try {
        one_step();
      } catch (...) {
        hipGraph_t g2 = nullptr;
        (void)hipStreamEndCapture(s, &g2);
        if (g2) (void)hipGraphDestroy(g2);
        throw;
      }
      SF_HIP(hipStreamEndCapture(s, &graph));
      if (h->gexec) {
        (void)hipGraphExecDestroy(h->gexec);
        h->gexec = nullptr;
      }
      ++h->graph_captures;
      hipError_t e = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) fail(SF_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      h->gkey = key;
      for (int i = 1; i < T; ++i) SF_HIP(hipGraphLaunch(h->gexec, s));
    } else {
      for (int i = 1; i < T; ++i) one_step();
    }
  }
  }
  SF_HIP(hipMemcpyAsync(x, xs, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (s != user) {
    SF_HIP(hipEventRecord(h->ev_out, s));
    SF_HIP(hipStreamWaitEvent(user, h->ev_out, 0));
  }
  fail_guard.armed = false;
  return SF_OK;
  SF_API_END
}

int sf_unet_debug_enable(sf_unet *h, float *buf, int64_t cap_floats) {
  if (!h) return SF_ERR_INVALID;
  h->dbg.buf = buf;
  h->dbg.cap = cap_floats;
  h->dbg.reset();
  return SF_OK;
}
int sf_unet_debug_count(const sf_unet *h) { return h ? (int)h->dbg.entries.size() : -1; }
int sf_unet_debug_info(const sf_unet *h, int i, char *name_out, int name_cap, int64_t *offset, int64_t *rows, int32_t *cols) {
  if (!h || i < 0 || i >= (int)h->dbg.entries.size()) return SF_ERR_INVALID;
  const auto &e = h->dbg.entries[i];
  if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", e.name.c_str());
  if (offset) *offset = e.offset;
  if (rows) *rows = e.rows;
  if (cols) *cols = e.cols;
  return SF_OK;
}
int sf_unet_launch_count(const sf_unet *h) { return h ? h->launches : -1; }
int sf_unet_graph_captures(const sf_unet *h) { return h ? h->graph_captures : -1; }

int sf_unet_set_branches(sf_unet *h, int n) {
  if (!h || n < 0 || n > sf_unet::kMaxBranches) return SF_ERR_INVALID;
  h->branches_override = n;
  return SF_OK;
}

int sf_unet_profile_enable(sf_unet *h, int on) {
  if (!h) return SF_ERR_INVALID;
  h->prof_on = on != 0;
  h->prof.clear();
  return SF_OK;
}
int sf_unet_profile_count(const sf_unet *h) { return h ? (int)h->prof.size() : -1; }
int sf_unet_profile_depth(const sf_unet *h, int i) {
  if (!h || i < 0 || i >= (int)h->prof.size()) return -2;
  return h->prof[i].depth;
}
int sf_unet_profile_get(const sf_unet *h, int i, char *name_out, int name_cap, float *ms, double *flops, double *bytes) {
  if (!h || i < 0 || i >= (int)h->prof.size()) return SF_ERR_INVALID;
  const auto &r = h->prof[i];
  if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", r.label);
  if (ms) *ms = r.ms;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return SF_OK;
}

}  // extern "C"
