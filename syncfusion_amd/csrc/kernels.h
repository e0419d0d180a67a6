// Host-side launchers of the gfx950 kernels.  Everything is channels-last:
// a 1-D activation is rows x channels with row = b * L + l; a video activation is
// rows = ((n * T + t) * H + h) * W + w.  `DT` selects fp32 (parity path) or bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

namespace sf {

// Tuning / measurement hooks (tools/, profiles/HISTORY*.md): environment variables that force tile variants, thresholds or drop launches.
// They exist ONLY in builds made with -DSF_TUNING_HOOKS (make tuning -> lib/libsyncfusion_amd_tuning.so, loaded through SF_LIB_PATH by
// the A/B scripts); the product library compiles every one of them to "not set" and contains none of the names.
#ifdef SF_TUNING_HOOKS
inline const char *tune_env(const char *name) { return getenv(name); }
#else
inline const char *tune_env(const char *) { return nullptr; }
#endif

// F32X (the channel-block chain only: conv_cb_shape_ok / launch_pack_conv_cb / launch_conv_cb): fp32 activations, split fp16 weights and
// panel (common.h, X3P<X3_F16>); every other launcher of the fp32x engine takes F32 plus ConvGemmArgs::wx
enum DType { F32 = 0, BF16 = 1, F16 = 2, F32X = 3 };
// split modes of the fp32x GEMMs (common.h, X3P<MODE>): fp16 hi + 2048-scaled fp16 lo (forward), bf16 hi + bf16 lo (gradients)
constexpr int X3_F16 = 1, X3_BF16 = 2;
inline size_t dsize(int dt) { return (dt == F32 || dt == F32X) ? 4 : 2; }
// three-way dispatch on the arithmetic type: SF_DISPATCH_T(dt, f<T>(args)) evaluates f with T = float / bf16 / f16;
// SF_DISPATCH_STMT(dt, statement using T) is the statement form (kernel launches)
#define SF_DISPATCH_T(dt, EXPR_T)             \
  ((dt) == ::sf::F32    ? [&] { using T = float; return EXPR_T; }()   \
   : (dt) == ::sf::BF16 ? [&] { using T = ::sf::bf16; return EXPR_T; }() \
                        : [&] { using T = ::sf::f16; return EXPR_T; }())
#define SF_DISPATCH_STMT(dt, STMT)                                   \
  do {                                                               \
    if ((dt) == ::sf::F32) { using T = float; STMT; }                \
    else if ((dt) == ::sf::BF16) { using T = ::sf::bf16; STMT; }     \
    else { using T = ::sf::f16; STMT; }                              \
  } while (0)

// ---------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the matrix cores:  out[m][n] = epi( sum_k A(m,k) * W[n][k] )
//   A(m, tap*cin + ci)       = pro( src[row(m,tap)][ci] )       (0 outside the padding)
//   A(m, taps*cin + ci2)     = src2[m][ci2]                      (channel concat, taps==1)
// ---------------------------------------------------------------------------------------
// Weight prefetch hosted by a kernel that leaves CUs idle: `wgs` extra workgroups appended to the host's grid read [ptr, ptr + bytes)
// once (16 bytes per lane) and exit, so that the NEXT GEMM of the chain finds its weights in the Infinity Cache / L2 instead of HBM
// (the 430 MB of bf16 weights of a step do not fit the 256 MB Infinity Cache: every GEMM streams its weights cold otherwise).
struct Prefetch {
  const void *ptr = nullptr;
  unsigned bytes = 0;
  int wgs = 0;
};

struct ConvGemmArgs {
  Prefetch pf;   // hosted prefetch (conv_gemm_wp / conv_gemm_fast only; other kernels ignore it)
  const void *src = nullptr, *src2 = nullptr, *w = nullptr, *res = nullptr;
  const void *wfr = nullptr;    // optional: the same [N][K] matrix in MFMA fragment order [N / 32][K / 16][64][8] (conv_gemm_rs.hip)
  const float *w32 = nullptr;   // the same weights in fp32, [N][taps*C] (8-channel level: conv_d0.hip reads these as scalar operands)
  // fp32 launches only ("x3", the fp32x engine): the same [N][K] matrix as split fp16 operands, [N][K / 32][hi 32 | lo' 32] (launch_pack_wx;
  // the same bytes per row as fp32).  When set, the kernels that carry the split mode multiply fp32 activations against it with three
  // v_mfma_f32_32x32x16_f16 per product (common.h, x3_split); every other kernel ignores it and multiplies `w` in fp32.
  const void *wx = nullptr;
  const void *wfrx = nullptr;   // the same split weights in MFMA fragment order [N / 32][K / 16][hi | lo'][64][8] (conv_gemm_rs.hip, K <= 1280)
  int wx_mode = 1;              // 1: fp16 hi + 2048-scaled fp16 lo (forward passes); 2: bf16 hi + bf16 lo (gradients: no range restriction)
  // the first source is ALREADY split (mode 1): rows of [cin / 32][hi 32 | lo' 32] fp16 -- the same bytes per row as fp32 -- written by a
  // producer whose output only this GEMM reads (launch_gn_silu with xfmt): the kernel then spends no vector instruction on the operand.
  // Honoured by the macro-tile kernel only (conv_gemm_src_x3_ok).
  int src_x3 = 0;
  // hint: nothing else runs beside this launch (the training step's single stream, an engine without clip-parallel branches) -- tile
  // choices that rely on a second stream filling the CUs a launch leaves idle do not apply
  int solo = 0;
  void *out = nullptr;
  const float *bias = nullptr, *gamma = nullptr, *beta = nullptr, *stats = nullptr;
  const float *badd = nullptr, *bscale = nullptr;
  int M = 0, N = 0, K = 0;  // K = taps*cin + cin2 (weights row length)
  int cin = 0, cin2 = 0, src_ld = 0, src2_ld = 0, out_ld = 0, res_ld = 0, n_store = 0;
  int geom = 0;  // 0: 1-D, 1: video (T,H,W)
  // 1-D: source position p = l*stride + tap - pad, valid when 0 <= p < Lsrc*up, source row = p >> up_shift
  int Lout = 1, Lsrc = 1, taps = 1, stride = 1, pad = 0, up_shift = 0;
  // video
  int To = 1, Ho = 1, Wo = 1, Ti = 1, Hi = 1, Wi = 1, kt = 1, kh = 1, kw = 1, st = 1, sh = 1, sw = 1, pt = 0, ph = 0, pw = 0;
  // prologue: 0 none, 1 GroupNorm+SiLU from a partial-statistics slab [B][nch][G][2] (mean, M2 per chunk)
  int pro = 0, G = 1, nch = 1, chunk_rows = 1;
  float eps = 1e-5f;
  // epilogue: v = acc + bias; v *= bscale[b][n]; v += res[m][n]; v += badd[b][n]; v = act(v)
  //   act: 0 none, 1 relu, 2 gelu(erf), 3 silu(gelu(v))   out_f32: store fp32 whatever the compute type
  int badd_ld = 0, bscale_ld = 0, act = 0, out_f32 = 0;
  // Row-LayerNorm fusion (conv_gemm_fast / conv_gemm_wp, 32x32 tiles only):
  //   producer side: rowpart_out != nullptr -> the epilogue also writes, per output row and 32-column tile, the (mean, M2)
  //     of the STORED values: rowpart_out[(m * rowpart_nt + n_tile) * 2 + {0,1}]   (rowpart_nt = n_store / 32);
  //   consumer side (launch_conv_gemm_ln): the first source (cin channels, one tap) is read as
  //     LayerNorm_cin(src; ln_eps) * (1 + ln_ss[b][c]) + ln_ss[b][cin + c]  (ln_ss == nullptr: plain normalisation), the row
  //     statistics pooled from ln_part (ln_nt = cin / 32 partials per row); res_ln: the residual rows get the same transform.
  float *rowpart_out = nullptr;
  int rowpart_nt = 0;
  // GroupNorm partials of the STORED output for the channel-block convolution that consumes it (conv_gemm_wp, 32x32 tiles, 1-D,
  // Lout >= 32 so that a tile touches at most two clips): gnpart_out[((m_tile * (n_store / 32) + n_tile) * 2 + seg) * 2 + {0,1}] =
  // (sum, sum of squares) over the tile's rows of the clip of its first row (seg 0) and of the next clip (seg 1).
  float *gnpart_out = nullptr;
  //     ln_colsum != nullptr (plain normalisation, no second source): the source is multiplied RAW and the LayerNorm is
  //     applied to the accumulator instead,  rstd_m * (acc[m][n] - mean_m * ln_colsum[n]),  ln_colsum[n] = sum_k w[n][k].
  const float *ln_colsum = nullptr;
  const float *ln_part = nullptr, *ln_ss = nullptr;
  int ln_nt = 0, ln_ss_ld = 0, res_ln = 0;
  float ln_eps = 1e-5f;
  // the caller allows the row-LayerNorm fusion on the MACRO tiles too (row partials in their epilogue, LayerNorm on the accumulator).  The
  // engine never sets it: in the two-branch step it measured -3.5 ... +0.9 % (profiles/r5_c_ab_mt_ln.txt); sf_op_inject_prenorm_proj does,
  // so that the kernels stay parity-tested.  An explicit argument: no process-wide switch for the dispatch to disagree with itself about.
  int mt_ln = 0;
};
struct V2Plan {
  int variant = 2;  // 0: 128x128, 1: 128x64, 2: 64x64
};
bool conv_gemm_v2_plan(int dt, const ConvGemmArgs &a, V2Plan &pl);
const char *conv_gemm_v2_name(int dt, const V2Plan &pl);
hipError_t launch_conv_gemm_v2(int dt, const ConvGemmArgs &a, const V2Plan &pl, hipStream_t s);
hipError_t launch_conv_gemm(int dt, const ConvGemmArgs &a, hipStream_t s);
// macro-tile kernel (conv_gemm_mt.hip): 256x128 tiles, LDS-DMA ring; 16-bit types, long activations
bool conv_gemm_mt_ok(int dt, const ConvGemmArgs &a);
bool conv_gemm_prefers_mt(const ConvGemmArgs &a);
bool conv_gemm_mt_wanted(int dt, const ConvGemmArgs &a);   // eligible AND preferred (the fp32 rule differs: conv_gemm_mt.hip)
// true when launch_conv_gemm(F32, a) would run on a kernel that can read its first source pre-split (ConvGemmArgs::src_x3)
bool conv_gemm_src_x3_ok(const ConvGemmArgs &a);
hipError_t launch_conv_gemm_mt(int dt, const ConvGemmArgs &a, hipStream_t s);
const char *conv_gemm_mt_name(const ConvGemmArgs &a);   // label of the tile variant it picks (bf16 spelling)
// true when launch_conv_gemm would run `a` on a kernel that honours rowpart_out (fast / wp, 32x32 tiles; macro tiles with SF_MT_LN=1)
bool conv_gemm_emits_rowpart(int dt, const ConvGemmArgs &a);
// register-staged kernel (conv_gemm_rs.hip): 32x32 tiles, all operand fragments of a wave in flight, fragment-ordered weights
bool conv_gemm_rs_ok(int dt, const ConvGemmArgs &a);
bool conv_gemm_rs_rows_ok(int64_t rows, int N);   // few enough 32x32 tiles for the small-batch kernels (the rule launch_conv_gemm applies)
hipError_t launch_conv_gemm_rs(int dt, const ConvGemmArgs &a, hipStream_t s);
hipError_t launch_pack_wfr(int dt, const void *w /* [N][K], compute type */, int N, int K, void *out, hipStream_t s);
hipError_t launch_pack_wfrx(const float *w /* [N][K] fp32 */, int N, int K, void *out, hipStream_t s);   // split fragment order (ConvGemmArgs::wfrx)
// split-fp16 image of a packed fp32 [N][K] matrix (K % 32 == 0): out[n][k / 32][0][k % 32] = hi, [1][k % 32] = lo' (common.h, x3_split)
hipError_t launch_pack_wx(const float *w, int N, int K, void *out, hipStream_t s, int mode = 1);
// Conv1d weight (N, C, taps) fp32 -> out[n][tap * C + c] fp32 and the split image of the same matrix, one pass ((taps * C) % 32 == 0)
hipError_t launch_pack_conv_x(const float *w, int N, int C, int taps, float *out, void *outx, int mode, hipStream_t s);
// true when launch_conv_gemm would run `a` on the kernel that honours gnpart_out (wp, 32x32 tiles)
bool conv_gemm_emits_gnpart(int dt, const ConvGemmArgs &a);
// GEMM whose first source is LayerNorm-modulated on the fly from producer-side row partials (see ConvGemmArgs)
bool conv_gemm_ln_ok(int dt, const ConvGemmArgs &a);
hipError_t launch_conv_gemm_ln(int dt, const ConvGemmArgs &a, hipStream_t s);
const char *conv_gemm_ln_variant_name(int dt, const ConvGemmArgs &a);
// Tuning hook (sf_bench_conv1d only; not thread-safe): force the kernel family / tile of launch_conv_gemm.
//   path: 0 automatic, 1 classic (conv_gemm), 2 wave-split-K (sk / fast), 4 v2;  tile: -1 automatic else variant index;
//   sk: reserved (was the grid split-K factor of v2; 64 selects the 256-wide chunk variant of conv_gemm_fast).
struct ConvGemmForce {
  int path = 0, tile = -1, sk = -1;
};
extern ConvGemmForce g_conv_gemm_force;
// name of the tile variant launch_conv_gemm picks for these arguments (profiling labels)
const char *conv_gemm_variant_name(int dt, const ConvGemmArgs &a);
bool conv_gemm_reads_split_only(int dt, const ConvGemmArgs &a);   // with ConvGemmArgs::wx at hand, is ConvGemmArgs::w never read? (conv_gemm.hip)
// the label of the bf16 build with the type renamed when dt == F16 (interned string)
const char *label_for_dtype(int dt, const char *bf16_label);
// bytes of dynamic LDS the GN table needs is bounded; returns false when the shape is unsupported.
bool conv_gemm_supported(int dt, const ConvGemmArgs &a);

// ---------------------------------------------------------------------------------------
// Thin-level convolution (conv_thin.hip): <= 64 output channels on long sequences, one workgroup per `rw` consecutive
// positions of a clip.  out[b,l,:] = (W . [pro(src)[b, (l-1..l+1) >> up_shift, :]  |  src2[b,l,:]] + bias) * bscale[b] (+ res) (+ badd[b])
//   pro 1: SiLU(GroupNorm(src))  with statistics merged from stats_in [B][nch_in][G][2] (mean, M2 per chunk_in rows)
//   pro 2: LayerNorm_C(src; eps) * (1 + ss[b][c]) + ss[b][C + c]   (ss == nullptr: plain normalisation)
//   res_self: the residual is pro(src) itself (InjectChannels after Modulation);  stats_out (optional): (mean, M2) of the
//   stored output per (clip, workgroup, group) = a slab with nch = nchw, chunk_rows = rw for the next GroupNorm.
// Weights: the packed [N = C][K = taps*C + C2] matrix of the implicit-GEMM path (compute type).
// ---------------------------------------------------------------------------------------
struct ConvThinArgs {
  const void *src = nullptr, *src2 = nullptr, *w = nullptr, *res = nullptr;
  const float *w32 = nullptr;   // the same weights in fp32, [N][taps*C] (8-channel level: conv_d0.hip reads these as scalar operands)
  void *out = nullptr;
  const float *bias = nullptr, *gamma = nullptr, *beta = nullptr, *stats_in = nullptr, *ss = nullptr, *badd = nullptr, *bscale = nullptr;
  float *stats_out = nullptr;
  // L output positions per clip, read from Ls = L >> up_shift source rows (nearest-neighbour upsampling when up_shift > 0);
  // C channels per source row, N output channels, C2 channels of the concatenated second source; weights [N][taps*C + C2]
  int B = 0, L = 0, Ls = 0, up_shift = 0, C = 0, N = 0, C2 = 0, taps = 1;
  int src_ld = 0, src2_ld = 0, out_ld = 0, res_ld = 0, ss_ld = 0, badd_ld = 0, bscale_ld = 0;
  int pro = 0, res_self = 0, G = 1, nch_in = 1, chunk_in = 1;
  float eps = 1e-5f;
  int rw = 32, nchw = 1;
  int x3 = 0;   // fp32 launches: products from split fp16 operands (the fp32x engine)
};
struct ThinPlan {
  int rw = 32, nchw = 1;
};
// C = channels of the level whose positions are tiled (the OUTPUT level of a down / up convolution)
ThinPlan conv_thin_plan(int B, int L, int C);
// Item tail of a thin level in one launch (conv_thin.hip):
//   y = x + Conv3(SiLU(GroupNorm(h)));  m = LayerNorm_C(y; eps_ln) * (1 + ss[b][c]) + ss[b][C + c];  out = m + W3 . [m | ctx] + bias3 (+ badd[b])
struct ThinTailArgs {
  const void *h = nullptr, *x = nullptr, *ctx = nullptr, *w2 = nullptr, *w3 = nullptr;
  const float *w2_32 = nullptr, *w3_32 = nullptr;   // fp32 [C][3*C] and [C][C + c2real] (8-channel level, conv_d0.hip)
  int c2real = 0;
  void *out = nullptr;
  const float *bias2 = nullptr, *bias3 = nullptr, *gamma = nullptr, *beta = nullptr, *stats_in = nullptr, *ss = nullptr, *badd = nullptr;
  float *stats_out = nullptr;
  int B = 0, L = 0, C = 0, C2 = 0, ctx_ld = 0, ss_ld = 0, badd_ld = 0, G = 1, nch_in = 1, chunk_in = 1;
  float eps_gn = 1e-5f, eps_ln = 1e-6f;
  int rw = 32, nchw = 1;
  int x3 = 0;   // fp32 launches: products from split fp16 operands (the fp32x engine)
};
bool thin_tail_supported(int dt, const ThinTailArgs &a);
hipError_t launch_thin_tail(int dt, const ThinTailArgs &a, hipStream_t s);
bool conv_thin_supported(int dt, const ConvThinArgs &a);
hipError_t launch_conv_thin(int dt, const ConvThinArgs &a, hipStream_t s);
// The 8-channel level on the vector units, one position per lane (conv_d0.hip): launch_conv_thin / launch_thin_tail route
// C = N = 8 shapes here (same argument blocks, chunking and statistics layout); SF_NO_D0=1 keeps the MFMA formulation.
bool d0_enabled(int B, int L);
bool d0_conv_supported(const ConvThinArgs &a);
bool d0_tail_supported(const ThinTailArgs &a);
hipError_t launch_d0_conv(int dt, const ConvThinArgs &a, hipStream_t s);
hipError_t launch_d0_tail(int dt, const ThinTailArgs &a, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Channel-block split-K convolution for the deep levels at small batch (conv_cb.hip): k = 3, stride 1, padding 1, 16-bit types.
//   slab[cb][m][n] = sum_{tap, c in block cb} W[n][c][tap] * pro(src)[m + tap - 1][c]      (fp32, no bias; cb = 0 .. C / 128 - 1)
//   pro 1: SiLU(GroupNorm(src)) from the chunk sums stats [B][nch][G][2] = (sum, sum of squares) that cb_reduce_gn leaves
//          (nch <= 32 chunks per clip), applied while the (rows + 2) x 128 activation panel is staged in LDS
//   pro 2: the same from the TILE sums the producing GEMM's epilogue leaves (ConvGemmArgs::gnpart_out: per 32-row x 32-column
//          tile and clip segment); needs C / G >= 32 and at most 32 tiles per (clip, group)
// The launches that follow sum the slabs: cb_reduce_gn (+ bias -> 16-bit h and its GroupNorm chunk partials) and cb_reduce_ln
// (+ bias + residual, LayerNorm over the row, Modulation -> 16-bit m).
// ---------------------------------------------------------------------------------------
struct ConvCbArgs {
  Prefetch pf;
  const void *src = nullptr;   // (B * L, src_ld) activations
  const void *wp = nullptr;    // weights in fragment order (launch_pack_conv_cb)
  float *slab = nullptr;       // [C / 128][B * L][N]
  int B = 0, L = 0, C = 0, N = 0, src_ld = 0;
  int pro = 0, G = 1, nch = 1, chunk_rows = 1;
  const float *gamma = nullptr, *beta = nullptr, *stats = nullptr;
  float eps = 1e-5f;
  int kb = 1;   // 128-channel blocks per workgroup: 1, or 2 (slab[C / 256][M][N]: half the partial slabs, twice the weight stream per workgroup)
  // filled by the launcher: log2(C / 128), log2(C / G), ceil(2^32 / L)
  int log2S = 0, log2cpg = 7;
  unsigned magicL = 0;
};
struct CbGnPlan {
  int nch = 1, chunk_rows = 8;
};
bool conv_cb_shape_ok(int dt, int B, int L, int C, int N, int G);
bool conv_cb_tile_stats_ok(int L, int C, int G);   // pro 2 applicable
int conv_cb_mt(int M, int N, int C);            // 32-row tiles per workgroup the launcher picks
size_t conv_cb_weight_elems(int N, int C);
hipError_t launch_pack_conv_cb(int dt, const float *w /* (N, C, 3) fp32 */, int N, int C, void *out, hipStream_t s);
hipError_t launch_conv_cb(int dt, const ConvCbArgs &a, hipStream_t s);
CbGnPlan cb_gn_plan(int L);
hipError_t launch_cb_reduce_gn(int dt, const float *slab, int S, int B, int L, int N, const float *bias, void *out, int out_ld, int G, float *stats,
                               const CbGnPlan &gp, hipStream_t s, Prefetch pf = Prefetch());
hipError_t launch_cb_reduce_ln(int dt, const float *slab, int S, int B, int L, int C, const float *bias, const void *res, int res_ld, const float *ss,
                               int ss_ld, float eps, void *out, int out_ld, hipStream_t s, Prefetch pf = Prefetch());

// ---------------------------------------------------------------------------------------
// Direct (VALU) convolution for thin layers (Cin*taps small, N <= 32): one output row per thread.
// Same A/epilogue semantics as ConvGemmArgs (1-D geometry only); weights fp32 [N][taps*cin + cin2].
// ---------------------------------------------------------------------------------------
hipError_t launch_conv_direct(int dt_in, int dt_out, const ConvGemmArgs &a, hipStream_t s);

// GroupNorm partial statistics: x:(B*L, C) (row stride ld) -> slab [B][nch][G][2] = (mean, M2) per chunk of rows.
struct GnPlan {
  int nch = 1, chunk_rows = 1;
};
GnPlan gn_plan(int B, int L, int C);
hipError_t launch_gn_stats(int dt, const void *x, int ld, int B, int L, int C, int G, int nch, int chunk_rows, float *slab,
                           hipStream_t s);

// y = silu(GroupNorm_G(x; gamma, beta, eps)) materialised in one launch (statistics + apply, one workgroup per (clip, group))
// xfmt (fp32 only, C % 32 == 0, out_ld == C): the output rows are written as split fp16 operands [C / 32][hi 32 | lo' 32] for a GEMM that
// takes ConvGemmArgs::src_x3
hipError_t launch_gn_silu(int dt, const void *x, int ld, int B, int L, int C, int G, const float *gamma, const float *beta, float eps,
                          void *out, int out_ld, hipStream_t s, Prefetch pf = Prefetch(), bool xfmt = false);
hipError_t launch_gn_silu_ws(int dt, const void *x, int ld, int B, int L, int C, int G, const float *gamma, const float *beta, float eps, void *out,
                             int out_ld, float *slab, int64_t slab_floats, hipStream_t s);

// y = LN_C(x; eps) * (1 + scale[b][c]) + shift[b][c]   (ss == nullptr: plain normalise);  ss:(B, ss_ld) = [scale | shift]
hipError_t launch_ln_modulate(int dt, const void *x, int ld, const float *ss, int ss_ld, float eps, int B, int L, int C,
                              void *out, int out_ld, hipStream_t s, Prefetch pf = Prefetch(), bool xfmt = false /* as launch_gn_silu */);

// Multi-head softmax attention on packed projections.  q row stride ldq, k/v inside kv with row stride ldkv
// (k at column 0, v at column H*D).  out row stride ldo.
// x3 (dt == F32 only): the products from split fp16 operands (the fp32x engine)
hipError_t launch_attention(int dt, const void *q, int ldq, const void *kv, int ldkv, int B, int L, int H, int D,
                            void *out, int ldo, hipStream_t s, bool x3 = false, bool xfmt = false /* x3 only, ldo == H * D: output rows pre-split */);

// ---------------------------------------------------------------------------------------
// Element-wise / layout helpers
// ---------------------------------------------------------------------------------------
// channels-first fp32 (B, C, L) -> channels-last DT (B*L, ld) with zero padding of columns [C, ld)
// every cross-attention output projection of a call in one launch (misc.hip): out[b][out_off + n] = sum_k w[n][k] v_all[b][v_off + k] + bias[n]
struct CrossOutItem {
  const void *w;       // [N][ldw] in the compute type
  const float *bias;   // [N] or null
  int N, ldw, v_off, out_off;
};
hipError_t launch_cross_out_grouped(int dt, const CrossOutItem *items, const int2 *blocks /* (item, first column) per workgroup */, int nblocks,
                                    const void *v_all, int ldv, int Bt, int hd, float *out, int out_ld, hipStream_t s);
hipError_t launch_cf_to_cl(int dt, const float *x, int B, int C, int L, void *out, int ld, hipStream_t s);
// channels-last DT (B*L, ld) -> channels-first fp32 (B, C, L)
hipError_t launch_cl_to_cf(int dt, const void *x, int ld, int B, int C, int L, float *out, hipStream_t s);
// video (N,3,T,H,W) fp32 -> channels-last DT rows x ld (ld >= 3, zero padded)
hipError_t launch_video_to_cl(int dt, const float *x, int N, int C, int T, int H, int W, void *out, int ld, hipStream_t s);
// DT rows x ld -> fp32 rows x C (debug taps)
hipError_t launch_to_f32(int dt, const void *x, int ld, int64_t rows, int C, float *out, hipStream_t s);

// learned-Fourier time embedding input: out[b] = [sigma, sin(2 pi sigma w), cos(2 pi sigma w), 0...]  (B, ld) as DT.
// sigma is read from sig[b] when sig_idx == nullptr, else from sig[*sig_idx] for every b (graph replay).
hipError_t launch_time_fourier(int dt, const float *sig, const int *sig_idx, const float *w, int B, int half, void *out, int ld,
                               hipStream_t s);

// v-sampler step (in place on x, fp32):  v = v_uncond ? v_u + (v_c - v_u)*scale : v ;
//   x <- a1*(a0*x - b0*v) + b1*(b0*x + a0*v),  (a0,b0,a1,b1) = sched[*step_idx][0..3].
hipError_t launch_vsampler_update(float *x, const float *v, const float *v_uncond, float scale, const float *sched,
                                  const int *step_idx, int64_t n, hipStream_t s);
// cur[0..ld) = table[*step_idx][0..ld), then (*step_idx)++  (single workgroup; first kernel of a sampling step)
hipError_t launch_step_select(const float *table, int ld, int *step_idx, float *cur, hipStream_t s);
// one wave busy-waits for `microseconds` (tuning aid)
hipError_t launch_spin(double microseconds, hipStream_t s);
// one wave measures the shader clock over `microseconds` of wall time: out2 = (shader cycles, 100 MHz ticks)
hipError_t launch_clock_probe(double microseconds, unsigned long long *out2, hipStream_t s);
hipError_t launch_touch(const void *p, size_t bytes, int wgs, unsigned *sink, hipStream_t s);
// (*step_idx)++ -- its own 1-thread launch so that no kernel of a step races with the increment
hipError_t launch_step_advance(int *step_idx, hipStream_t s);
// out = v_u + (v_c - v_u) * scale   (single forward with CFG)
hipError_t launch_cfg_combine(const float *v_c, const float *v_u, float scale, float *out, int64_t n, hipStream_t s);

// mean over the spatial rows of a video activation: x:(N*T*HW, ld) -> out:(N*T, C) fp32
hipError_t launch_spatial_mean(int dt, const void *x, int ld, int NT, int HW, int C, float *out, hipStream_t s);

// polyphase sinc resampler (resample.hip): bank [nnew][2*width+orig] fp32 on the device; x:(R,L) -> out:(R,Lout)
hipError_t launch_resample(const float *x, int R, int L, const float *bank, int orig, int nnew, int width, float *out, int Lout,
                           hipStream_t s);

// cut_prefix + crop after sampling: first[b] = first non-zero of y[b,0,:] (L if none); out[b,c,l<Lc] = l < first[b] ? 0 : gen[b,c,l]
hipError_t launch_cut_prefix_crop(const float *gen, const float *y, int B, int C, int L, int Lc, float *out, int *first, hipStream_t s);
// onset glue (sf_onsets_to_track)
hipError_t launch_onsets_to_track(const float *logits, int N, int T, const int32_t *start_frame, float frame_rate,
                                  float sample_rate, float threshold, float *track, int L, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Weight packing (run once at engine creation)
// ---------------------------------------------------------------------------------------
// out[n] = sum_k w[n][k] of a packed [N][K] matrix (as stored, i.e. of the rounded compute-type values)
hipError_t launch_row_sums(int dt, const void *w, int N, int K, float *out, hipStream_t s);
// conv weight (N, Ctot, taps) fp32, channels [c_off, c_off+Cin) -> out[n*out_row + col0 + tap*cin_pad + ci] as DT,
// zero for ci in [Cin, cin_pad)  (+ optional per-N scale = folded BatchNorm)
hipError_t launch_pack_conv(int dt, const float *w, int N, int Ctot, int c_off, int Cin, int taps, int cin_pad, const float *nscale,
                            void *out, int64_t out_row, int64_t col0, hipStream_t s);
// ConvTranspose1d weight (Cin, Cout, f) fp32 (kernel = stride = f) -> [n = t*Cout + o][k = c] as DT, row length out_row >= Cin (zero padded)
hipError_t launch_pack_convT(int dt, const float *w, int Cin, int Cout, int f, void *out, int64_t out_row, hipStream_t s);
// generic strided copy/convert: out[r*ldo + c] = (DT) in[r*ldi + c] * (cscale ? cscale[c] : 1)
hipError_t launch_pack_rows(int dt, const float *in, int64_t rows, int cols, int64_t ldi, const float *cscale, void *out,
                            int64_t ldo, hipStream_t s);
// out[n] = sum_k w[n][k] * v[k] (+ add[n])   fp32 gemv used for folding LayerNorm biases at pack time
hipError_t launch_fold_bias(const float *w, int N, int K, const float *v, const float *add, float *out, hipStream_t s);
// ---------------------------------------------------------------------------------------
// VideoOnsetNet stem (onset_stem.hip): Conv3d(3 -> <= 64, (1,7,7), stride (1,2,2), pad (0,3,3)) + shift + ReLU, 16-bit types
// ---------------------------------------------------------------------------------------
size_t onset_stem_weight_elems();
// generic packed weights [n][49 taps][4] (row length k_in, compute type) -> the kernel's [64][7][8][4] image
hipError_t launch_onset_stem_repack(int dt, const void *w_generic, int n_real, int k_in, void *out, hipStream_t s);
hipError_t launch_onset_stem(int dt, const void *in, int NT, int H, int W, const void *wk, const float *shift, int n_real, void *out, int out_ld,
                             hipStream_t s);

// ---------------------------------------------------------------------------------------
// Temporal-walk (3,1,1) convolution, 64 output channels, 16-bit types (conv_tw.hip): a workgroup owns 128 positions of one clip and walks
// the frames with a three-frame LDS ring and register-stationary weights (VideoOnsetNet stem / layer 1).
// ---------------------------------------------------------------------------------------
bool conv_tw_ok(int dt, int cin_real, int cin_ld, int cout, int out_ld, int res_ld);
size_t conv_tw_weight_elems(int cin_real);
hipError_t launch_pack_conv_tw(int dt, const void *w /* [64][3 * cin_ld], compute type */, int cin_real, int cin_ld, void *out, hipStream_t s);
hipError_t launch_conv_tw(int dt, const void *in, int in_ld, int cin_real, const void *wfr, const float *bias, const void *res, int res_ld, void *out,
                          int out_ld, int N, int T, int HW, int relu, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Frame-walk (1,3,3) convolution, 64 input channels, stride 1, 16-bit types (conv_sp.hip): a workgroup owns one 32-column tile of the output and
// an 8 x 14 patch of positions and walks the frames with register-stationary weights and one LDS halo tile per frame (VideoOnsetNet layer 1).
// ---------------------------------------------------------------------------------------
bool conv_sp_ok(int dt, int cin_real, int cin_ld, int cout, int out_ld);
size_t conv_sp_weight_elems(int cout);
hipError_t launch_pack_conv_sp(int dt, const void *w /* [cout][9 * 64], compute type */, int cout, void *out, hipStream_t s);
hipError_t launch_conv_sp(int dt, const void *in, int in_ld, const void *wfr, const float *shift, int cout, void *out, int out_ld, int N, int T, int H,
                          int W, int relu, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Input side (input.hip)
// ---------------------------------------------------------------------------------------
// uint8 RGB frames (N, T, H, W, 3) -> (N, 3, T, oh, ow) fp32: /255, antialiased bilinear resize (ATen semantics), (x - mean) / std
hipError_t launch_frames_preprocess(const unsigned char *frames, int N, int T, int H, int W, int oh, int ow, const float *mean, const float *stdv,
                                    float *out, hipStream_t s);
// track (B, L) = 0 except track[clip_of[i]][int(times[i] * sample_rate)] = 1
hipError_t launch_times_to_track(const double *times, const int *clip_of, int n_times, double sample_rate, int B, int L, float *track, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Training backward, first slice (train.hip): fp32, channels-last
// ---------------------------------------------------------------------------------------
// Conv1d weight (N, C, taps) -> dgrad matrix [c][t' * ldn + n] = W[n][c][taps-1-t'] (the forward kernels then compute da from dy)
// outx (optional, (taps * ldn) % 32 == 0): the split bf16 image of the same matrix, written in the same pass
hipError_t launch_pack_dgrad(const float *w, int N, int C, int taps, int ldn, float *out, hipStream_t s, void *outx = nullptr);
// every weight image of one training convolution in one launch (train.hip): fw [N][taps][C] fp32, fwx its split fp16 image, dg the
// data-gradient matrix [C][taps][N] (taps flipped) fp32, dgx its split bf16 image; null outputs are skipped; taps <= 9
hipError_t launch_pack_train(const float *w, int N, int C, int taps, float *fw, void *fwx, float *dg, void *dgx, hipStream_t s);
// the same for many weights in one launch: desc_dev = n_items x 7 64-bit words in device memory (w, fw, fwx, dg, dgx, N | C << 32,
// taps | first_tile << 32), items sorted by first_tile, tiles of an item = ceil(C / 32) * ceil(N / 32), total_tiles their sum
hipError_t launch_pack_train_many(const void *desc_dev, int n_items, int total_tiles, hipStream_t s);
// dw (N, C, taps) = sum_rows dy[row][n] * act[row + t - pad][c];  partial: [S][N][taps*C] scratch, S = conv_wgrad_splits(...)
int conv_wgrad_splits(int64_t rows, int C, int N, int taps);
// x3: the products from split fp16 operands (both operands are activations: split while they are staged)
// (bias_part / bias_slices / db: the bias gradient's slice sums, written by launch_col_sums_part BEFORE this call, are reduced by workgroups
//  appended to the weight gradient's reducer; *bias_done tells whether that happened -- not with one row split or C % 4 != 0)
hipError_t launch_conv_wgrad(const float *dy, const float *act, int B, int L, int C, int N, int taps, int pad, float *partial, int S, float *dw,
                             hipStream_t s, int x3_mode = 0, const float *bias_part = nullptr, int bias_slices = 0, float *db = nullptr,
                             bool *bias_done = nullptr);
hipError_t launch_col_sums_part(const float *x, int64_t rows, int cols, float *part, int S, hipStream_t s);
hipError_t launch_slices_reduce(const float *part, int S, int cols, float *out, hipStream_t s);
// out[col] = sum_rows x[row][col]   (part: [S][cols] scratch)
hipError_t launch_col_sums(const float *x, int64_t rows, int cols, float *part, int S, float *out, hipStream_t s);
// out[b][c] = sum_l x[b][l][c] * (y ? y[b][l][c] : 1); part: B * length_sums_slices(B, L) * C floats
int length_sums_slices(int B, int L);
hipError_t launch_length_sums(const float *x, const float *y, int B, int L, int C, float *part, float *out, hipStream_t s);
// backward of a = SiLU(GroupNorm_G(x; gamma, beta, eps)): dx, and dgb = [dgamma | dbeta]  (dgb_part: [B][2][C] scratch)
hipError_t launch_gn_silu_bwd(const float *x, const float *da, const float *gamma, const float *beta, int B, int L, int C, int G, float eps,
                              float *dx, float *dgb_part, float *dgb, hipStream_t s, const float *slab_in = nullptr,
                              const float *dx_add = nullptr /* added to dx in the same pass: a residual branch's gradient of x */);
int64_t gn_bwd_stats_floats(int B, int L, int C, int G);
int64_t gn_silu_bwd_ws_floats(int B, int L, int C, int G);
hipError_t launch_gn_silu_recompute(const float *x, const float *gamma, const float *beta, int B, int L, int C, int G, float eps, float *act,
                                    float *ws, hipStream_t s);
hipError_t launch_gn_bwd_stats(const float *x, int B, int L, int C, int G, float *ws, hipStream_t s);

// backward of y = LayerNorm_C(x; eps) * (1 + ss[b][c]) + ss[b][C + c] (ss == nullptr: plain normalisation): dx and, when dss != nullptr,
// dss (B, 2C) = [dscale | dshift];  dss_part: [B][ln_mod_bwd_chunks(L, C)][2C] scratch
int ln_mod_bwd_chunks(int L, int C);
hipError_t launch_ln_modulate_bwd(const float *x, const float *ss, const float *dy, float eps, int B, int L, int C, float *dx, float *dss_part,
                                  float *dss, hipStream_t s, const float *dx_add = nullptr /* added to dx in the same pass */);
// backward of softmax attention on packed projections (head dim 64): dq (B,L,H*64), dkv (B,L,2*H*64); lse, dsum: (B,H,L) scratch
hipError_t launch_attention_bwd(const float *q, const float *kv, const float *o, const float *dout, int B, int L, int H, int D, float *dq, float *dkv,
                                float *lse, float *dsum, hipStream_t s, const float *lse_fwd = nullptr, bool x3 = false);
// fp32 attention forward on the matrix cores that also keeps log-sum-exp of the scaled scores, (B, H, L), for launch_attention_bwd(lse_fwd)
bool attention_f32_mfma_ok(int ldq, int ldkv, int ldo, int B, int H);
hipError_t launch_attention_f32_mfma(const float *q, int ldq, const float *kv, int ldkv, int B, int L, int H, float *out, int ldo, hipStream_t s,
                                     float *lse_out, bool x3 = false, bool xfmt = false);

// BatchNorm (eval) -> per-channel scale / shift
hipError_t launch_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int C,
                          float *scale, float *shift, hipStream_t s);

}  // namespace sf
