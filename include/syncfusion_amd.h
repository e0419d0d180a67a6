/*
 * syncfusion_amd -- C ABI of the MI355X (gfx950) implementation of SyncFusion's
 * generation hot path.
 *
 * The reference (mcomunita/syncfusion) has no FFI: its "plugin API" is Hydra
 * `_target_` instantiation plus Python duck typing.  The
 * entry points below are what a binding for that path would bind; each cites the
 * reference interface it replaces.  All pointers are DEVICE pointers unless
 * marked host; all tensors are fp32, contiguous, in the reference's own
 * (channels-first PyTorch) layout at the boundary.  No ownership is transferred,
 * no exceptions cross the ABI, every call returns SF_OK (0) or an SF_ERR_* code
 * and leaves a message in sf_last_error().  `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  Nothing here allocates inside the timed path:
 * activations live in a caller-provided workspace.
 */
#ifndef SYNCFUSION_AMD_H
#define SYNCFUSION_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_MAX_DEPTH 12

enum {
  SF_OK = 0,
  SF_ERR_INVALID = 1,        /* bad argument / inconsistent config                       */
  SF_ERR_MISSING_WEIGHT = 2, /* a named parameter was not supplied or has the wrong size  */
  SF_ERR_SHAPE = 3,          /* B/L0/T/H/W not supported by this model (stride, length)   */
  SF_ERR_HIP = 4,            /* a HIP runtime call failed                                 */
  SF_ERR_WORKSPACE = 5,      /* workspace missing or too small                            */
  SF_ERR_UNSUPPORTED = 6
};

enum { SF_UP_NEAREST_CONV3 = 0, SF_UP_TRANSPOSE = 1 };
/* arithmetic / storage type of the activations and packed weights (accumulation, statistics, softmax and the sampler state are fp32 in every
 * mode).  SF_F32X ("fp32x", the parity-grade fast path): activations stay fp32 in HBM, every matrix product is built from split fp16
 * operands -- a = hi + lo'/2048, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation -- which measures 7.5e-8 rel-L2 against fp64
 * on a K = 3072 GEMM (plain fp32 MFMA: 3.5e-7) at several times the fp32 matrix rate; operands must stay inside the fp16 range (< 65504). */
enum { SF_F32 = 0, SF_BF16 = 1, SF_F16 = 2, SF_F32X = 3 };

/* One named parameter of a torch state_dict: fp32, contiguous, PyTorch layout, device memory. */
typedef struct {
  const char *name;
  const void *data;
  int64_t numel;
} sf_tensor;

const char *sf_version(void);
const char *sf_last_error(void);
/* 1 when a gfx950 device is visible to the HIP runtime, else 0 (never fails). */
int sf_device_ok(void);
/* Measurement aid (bench.py; no reference counterpart): the shader clock the chip holds WHILE a workload runs.  _start puts a one-wave
 * kernel on `stream` (a side stream) that compares the shader-cycle counter with the constant 100 MHz counter for `microseconds`;
 * _read waits for it and returns MHz (host pointer).  One probe at a time per process. */
int sf_clock_probe_start(double microseconds, void *stream);
int sf_clock_probe_read(double *mhz_out);

/* ------------------------------------------------------------------------------------------
 * U-Net denoiser + v-sampler
 *   replaces: audio_diffusion_pytorch.UNetV0 / DiffusionModel.sample as instantiated by
 *   exp/model/diffusion.yaml:11-33 and called from main/generation.py:77-83,
 *   main/module_diffusion.py:77,200-206.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_layers;
  int32_t in_channels;
  int32_t channels[SF_MAX_DEPTH];
  int32_t factors[SF_MAX_DEPTH];
  int32_t items[SF_MAX_DEPTH];
  int32_t attentions[SF_MAX_DEPTH];
  int32_t cross_attentions[SF_MAX_DEPTH];
  int32_t context_channels[SF_MAX_DEPTH];
  int32_t attention_heads;
  int32_t attention_features;
  int32_t embedding_features;
  int32_t embedding_max_length;
  int32_t modulation_features;
  int32_t resnet_groups;
  int32_t dtype; /* SF_F32 (parity path), SF_BF16 or SF_F16 */
  /* Up path of a block (a-unet apex.py): SF_UP_NEAREST_CONV3 = nn.Upsample(nearest) + Conv1d(k=3) [UpsampleInterpolate];
   * SF_UP_TRANSPOSE = ConvTranspose1d(kernel = stride = factor) [Upsample]; `blocks.{d}.up.weight` is (in, C, 3) resp. (C, in, factor). */
  int32_t upsample_mode;
  /* [RECALLED] facts about a-unet's TimeConditioningPlugin / AttentionBase that only the upstream package can settle;
   * zero = SURVEY appendix A.3 as written.  tools/pin_upstream.py decides them, keymap.py reads the first and the last off a checkpoint.
   *   time_fourier_features: learned frequencies of the time embedder (0 = modulation_features / 2; a-unet NumberEmbedder(dim=256) = 128):
   *     `time.fourier_w` has this many entries, `time.lin0.weight` is (modulation_features, 1 + 2 * time_fourier_features);
   *   time_no_first_act: 1 = no GELU between the embedder's Linear and the two (Linear, GELU) layers;
   *   attention_out_bias: 1 = every `to_out` Linear of the attention items carries a bias (`<item>.{attn,cross}.to_out.bias`). */
  int32_t time_fourier_features, time_no_first_act, attention_out_bias;
} sf_unet_config;

typedef struct sf_unet sf_unet;

/* Build the engine: packs/folds the named fp32 parameters into its own device buffers
 * (weights are copied; the caller may free `weights` afterwards).  Parameter names are
 * listed by sf_unet_param_name(). */
int sf_unet_create(const sf_unet_config *cfg, const sf_tensor *weights, int n_weights, void *stream, sf_unet **out);
void sf_unet_destroy(sf_unet *h);

/* Enumerate the parameter names/sizes the engine expects for `cfg` (host strings). */
int sf_unet_param_count(const sf_unet_config *cfg);
int sf_unet_param_name(const sf_unet_config *cfg, int index, char *name_out, int name_cap, int64_t *numel_out);

/* Bytes of caller-allocated workspace needed for batch B and length L0 (two_pass != 0 when
 * embedding_scale != 1: cond and uncond evaluations run as one 2B batch). */
int64_t sf_unet_workspace_bytes(const sf_unet *h, int B, int L0, int two_pass);
/* Same for sf_vsample with `num_steps` steps (adds the sigma schedule and the per-step modulation table). */
int64_t sf_vsample_workspace_bytes(const sf_unet *h, int B, int L0, int two_pass, int num_steps);

/* One denoiser evaluation  v = net(x, sigma; channels, embedding, embedding_scale)
 *   (replaces UNetV0.forward, reached from main/module_diffusion.py:77 through VDiffusion).
 *   x, out: (B, in_channels, L0).  sigma: (B).  ctx[d]: (B, context_channels[d], L0/prod(factors[:d+1])).
 *   emb: (B, embedding_max_length, embedding_features). */
int sf_unet_forward(sf_unet *h, const float *x, const float *sigma, const float *const *ctx, const float *emb,
                    int B, int L0, float embedding_scale, float *out, void *ws, int64_t ws_bytes, void *stream);

/* The whole sampling loop  x <- VSampler(net)(x_noisy, num_steps, ...)  in place
 *   (replaces DiffusionModel.sample, main/generation.py:77-83, main/module_diffusion.py:200-206).
 *   use_graph != 0 captures one step into a hipGraph and replays it. */
int sf_vsample(sf_unet *h, float *x_inout, const float *const *ctx, const float *emb, int B, int L0,
               int num_steps, float embedding_scale, int use_graph, void *ws, int64_t ws_bytes, void *stream);

/* Debug taps (tests only): after the next sf_unet_forward, every block-level activation is
 * copied as fp32 channels-last rows into `buf` (device).  Query the table afterwards. */
int sf_unet_debug_enable(sf_unet *h, float *buf, int64_t cap_floats);
int sf_unet_debug_count(const sf_unet *h);
int sf_unet_debug_info(const sf_unet *h, int i, char *name_out, int name_cap, int64_t *offset, int64_t *rows, int32_t *cols);

/* Per-kernel launch accounting of the last forward (host side, for bench.py's roofline):
 * number of kernel launches in one evaluation. */
int sf_unet_launch_count(const sf_unet *h);
/* Step graphs captured and instantiated by sf_vsample so far.  Graphs are cached per (shape, workspace, guidance) -- the active set
 * plus three stashed ones -- so a server alternating between a few request shapes stops adding to this after its first round. */
int sf_unet_graph_captures(const sf_unet *h);
/* Number of clip-parallel branches (independent slices of the batch run concurrently on separate HIP streams,
 * forked/joined with events): 0 = automatic (2 for >= 4 clips), 1 = off, up to 8. */
int sf_unet_set_branches(sf_unet *h, int n);
/* Per-launch timing of the next sf_unet_forward: HIP events recorded on `stream` around every kernel launch
 * (label = kernel / tile variant; flops, bytes = ALGORITHMIC work of that launch). */
int sf_unet_profile_enable(sf_unet *h, int on);
int sf_unet_profile_count(const sf_unet *h);
int sf_unet_profile_get(const sf_unet *h, int i, char *name_out, int name_cap, float *ms, double *flops, double *bytes);
/* U-Net depth (block index, 0 = outermost) of profile record i; -1 for per-step features, -2 for a bad index.  Feeds the
 * per-depth-group roofline SURVEY.md section 8d asks for (HBM side for depths 0-3, MFMA side for depths 4-7). */
int sf_unet_profile_depth(const sf_unet *h, int i);

/* ------------------------------------------------------------------------------------------
 * Encoder1d (onset-track feature pyramid)
 *   replaces: audio_encoders_pytorch.Encoder1d as instantiated by exp/model/diffusion.yaml:35-43
 *   and called as onsets_encoder(y, with_info=True) at main/generation.py:71,
 *   main/module_diffusion.py:76,196.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_layers; /* len(factors) */
  int32_t in_channels;
  int32_t channels;
  int32_t multipliers[SF_MAX_DEPTH + 1];
  int32_t factors[SF_MAX_DEPTH];
  int32_t num_blocks[SF_MAX_DEPTH];
  int32_t resnet_groups;
  int32_t patch_size; /* must be 1 */
} sf_encoder1d_config;

typedef struct sf_encoder1d sf_encoder1d;

int sf_encoder1d_create(const sf_encoder1d_config *cfg, const sf_tensor *weights, int n_weights, void *stream, sf_encoder1d **out);
void sf_encoder1d_destroy(sf_encoder1d *h);
int64_t sf_encoder1d_workspace_bytes(const sf_encoder1d *h, int B, int L0);
/* y: (B, in_channels, L0).  xs_out[0] = to_in(y), xs_out[1+i] = downsample_i output, each
 * (B, C_i, L_i) fp32 channels-first: n_layers + 1 pointers (info["xs"][1:-1] of the reference). */
int sf_encoder1d_forward(sf_encoder1d *h, const float *y, int B, int L0, float *const *xs_out,
                         void *ws, int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * VideoOnsetNet (R(2+1)D-18 with the temporal stride removed + FC head)
 *   replaces: main/onset_net.py:46-63 (VideoOnsetNet.forward), main/resnet.py:234-251.
 *   Weights use the reference's own state_dict names (net.model.stem.0.weight ... fc.2.bias),
 *   BatchNorm in eval mode (running statistics are folded into the convolutions).
 * ---------------------------------------------------------------------------------------- */
typedef struct sf_onsetnet sf_onsetnet;

int sf_onsetnet_create(const sf_tensor *weights, int n_weights, int dtype, void *stream, sf_onsetnet **out);
void sf_onsetnet_destroy(sf_onsetnet *h);
int64_t sf_onsetnet_workspace_bytes(const sf_onsetnet *h, int N, int T, int H, int W);
/* frames: (N, 3, T, H, W); logits: (N, T) raw (no sigmoid), as the reference returns them. */
int sf_onsetnet_forward(sf_onsetnet *h, const float *frames, int N, int T, int H, int W, float *logits,
                        void *ws, int64_t ws_bytes, void *stream);
int sf_onsetnet_debug_enable(sf_onsetnet *h, float *buf, int64_t cap_floats);
int sf_onsetnet_debug_count(const sf_onsetnet *h);
int sf_onsetnet_debug_info(const sf_onsetnet *h, int i, char *name_out, int name_cap, int64_t *offset, int64_t *rows, int32_t *cols);

/* ------------------------------------------------------------------------------------------
 * Onset glue on device: logits -> one-hot impulse track
 *   replaces: main/module_onset.py:160-183 (threshold raw logits > 0.5, t = (idx+start)/fps,
 *   "%.4f" rounding) + main/dataset_diffusion.py:69-72 (track[:, int(t*sr)] = 1).
 *   logits: (N, T); track: (N, 1, L) is zero-filled then set. */
int sf_onsets_to_track(const float *logits, int N, int T, const int32_t *start_frame /* (N) or NULL */,
                       float frame_rate, float sample_rate, float threshold, float *track, int L, void *stream);

/* ------------------------------------------------------------------------------------------
 * Post-sampling cut_prefix + crop on device
 *   replaces: main/generation.py:86-89 (gen[i, :, :nonzero(y[i][0])[0]] = 0) and :100 (gen[..., :cut_length]).
 *   gen: (B, C, L); y: (B, 1, L) impulse track; out: (B, C, cut_length); first_onset: (B) int32 device array that
 *   receives the index of the first onset of every clip, L when the track is empty (the reference raises IndexError
 *   there: the caller checks). */
int sf_cut_prefix_crop(const float *gen, const float *y, int B, int C, int L, int cut_length, float *out, int32_t *first_onset,
                       void *stream);

/* ------------------------------------------------------------------------------------------
 * Post-sampling resampler
 *   replaces: torchaudio.functional.resample(gen[i, :, :cut_length].cpu(), orig_freq=sample_rate,
 *   new_freq=downsample_rate) at main/generation.py:91-98 (torchaudio==0.13.1 defaults: windowed-sinc, Hann,
 *   lowpass_filter_width 6, rolloff 0.99).  x: (R, L) fp32 rows; out: (R, ceil(new*L/orig)).
 * ---------------------------------------------------------------------------------------- */
typedef struct sf_resampler sf_resampler;
int sf_resampler_create(int orig_freq, int new_freq, int lowpass_filter_width, float rolloff, sf_resampler **out);
void sf_resampler_destroy(sf_resampler *h);
int sf_resampler_out_length(const sf_resampler *h, int L);
int sf_resampler_forward(sf_resampler *h, const float *x, int R, int L, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Input side
 *   sf_frames_preprocess replaces the per-frame transform of main/dataset_onset.py:47-50,152-165
 *     (ToTensor -> Resize((112,112), antialias=True) -> Normalize(mean, std) -> (C,T,H,W)) for a batch of decoded frames:
 *     frames:(N,T,H,W,3) uint8 device memory -> out:(N,3,T,out_h,out_w) fp32; mean3 / std3 are HOST arrays of 3 floats.
 *   sf_times_to_track replaces the impulse-track construction of main/dataset_diffusion.py:58-72
 *     (`onset[:, int(t * sr)] = 1.0`): times:(n_times) float64 seconds and clip_of:(n_times) clip indices, device memory.
 * ---------------------------------------------------------------------------------------- */
int sf_frames_preprocess(const uint8_t *frames, int N, int T, int H, int W, int out_h, int out_w, const float *mean3, const float *std3,
                         float *out, void *stream);
int sf_times_to_track(const double *times, const int32_t *clip_of, int n_times, double sample_rate, int B, int L, float *track, void *stream);

/* ------------------------------------------------------------------------------------------
 * Op-level entry points (channels-last, used by tests/ to localise kernel bugs).
 *   dtype selects the storage type of x / w / out (fp32 or bf16 bit patterns).
 * ---------------------------------------------------------------------------------------- */
/* out = conv1d(silu(groupnorm(x))) + bias (+ residual).  x:(B,L,C) and out/residual:(B,Lout,N) channels-last in
 * `dtype`; w:(N,C,taps) fp32 in PyTorch layout (packed internally); groups==0 -> no norm/activation;
 * upsample = nearest-neighbour factor applied before the convolution.  Lout = (L*upsample + 2*pad - taps)/stride + 1. */
int sf_op_conv1d_cl(int dtype, const void *x, const float *w, const float *bias, const float *gamma, const float *beta,
                    int groups, float eps, const void *residual, int B, int L, int C, int N, int taps, int stride,
                    int pad, int upsample, void *out, void *ws, int64_t ws_bytes, void *stream);
/* Training backward, first slice (the reference's training step is main/module_diffusion.py:73-82 under
 * exp/train_diffusion_gh.yaml:84-96, fp32): gradients of  y = conv1d(act(x)) + bias  with  act = silu(groupnorm(x)) when
 * groups > 0 (a ResnetItem convolution) or the identity when groups == 0 (the 1x1 InjectChannels convolution), stride 1,
 * 2 * pad == taps - 1, everything fp32 channels-last: x, dx:(B,L,C); dy:(B,L,N); w, dw:(N,C,taps) PyTorch layout; db:(N) or NULL;
 * dgb:(2C) = [dgamma | dbeta] (groups > 0).  dx may be NULL when groups == 0 and dw may be NULL (the caller needs no such gradient:
 * that part of the work is skipped).  No atomics: results are bit-reproducible. */
int64_t sf_op_conv1d_bwd_workspace_bytes(int B, int L, int C, int N, int taps, int groups);
int sf_op_conv1d_bwd_cl(const float *x, const float *w, const float *gamma, const float *beta, int groups, float eps, const float *dy,
                        int B, int L, int C, int N, int taps, int pad, float *dx, float *dw, float *db, float *dgb, void *ws,
                        int64_t ws_bytes, void *stream);
/* The training forward of the GroupNorm convolutions keeps a = silu(groupnorm(x)) and the chunk statistics the GroupNorm backward reads:
 *   sf_op_gn_silu_train: act:(B,L,C) fp32, stats: sf_op_gn_silu_train_stats_floats() floats (0: none are kept for this shape, stats may be NULL);
 *   sf_op_conv1d_bwd_cl_act = sf_op_conv1d_bwd_cl with act (and optionally stats) handed in: nothing is recomputed. */
int64_t sf_op_gn_silu_train_stats_floats(int B, int L, int C, int groups);
int sf_op_gn_silu_train(const float *x, const float *gamma, const float *beta, int groups, float eps, int B, int L, int C, float *act, float *stats,
                        void *stream);
int sf_op_conv1d_bwd_cl_act(const float *x, const float *act, const float *stats /* or NULL */, const float *w, const float *gamma, const float *beta,
                            int groups, float eps, const float *dy, int B, int L, int C, int N, int taps, int pad, float *dx, float *dw, float *db,
                            float *dgb, void *ws, int64_t ws_bytes, void *stream);
/* The same with the arithmetic of the two GEMMs chosen by `dtype`: SF_F32 (v_mfma_f32_32x32x2_f32) or SF_F32X (fp32 tensors, products from
 * split fp16 operands: data gradient through the forward kernels' split mode, weight gradient with both operands split while staged).
 * act / stats may be NULL (recomputed). */
int sf_op_conv1d_bwd_cl_x(int dtype, const float *x, const float *act, const float *stats, const float *w, const float *gamma, const float *beta,
                          int groups, float eps, const float *dy, int B, int L, int C, int N, int taps, int pad, float *dx, float *dw, float *db,
                          float *dgb, void *ws, int64_t ws_bytes, void *stream);
/* One weight-pack launch per convolution and training step.  sf_op_conv1d_train_fwd = sf_op_conv1d_cl for fp32 tensors (dtype SF_F32 /
 * SF_F32X, stride 1, no upsampling, taps <= 9) that ALSO writes the images of `w` the data-gradient GEMM of the backward pass reads into
 * dgrad_pack (>= sf_op_conv1d_dgrad_pack_bytes(C, N, taps) bytes: the flipped / transposed matrix [C][taps][N] in fp32, then its split
 * bf16 image); dgrad_pack may be NULL (then it is sf_op_conv1d_cl).  sf_op_conv1d_bwd_cl_p = sf_op_conv1d_bwd_cl_x reading those
 * images instead of packing its own (dgrad_pack NULL: packs its own).  The caller keeps dgrad_pack alive and `w` unchanged between the
 * two calls (the reference's training step: main/module_diffusion.py:79-82, optimizer step after backward). */
int64_t sf_op_conv1d_dgrad_pack_bytes(int C, int N, int taps);
int sf_op_conv1d_train_fwd(int dtype, const float *x, const float *w, const float *bias, const float *gamma, const float *beta, int groups, float eps,
                           const float *residual, int B, int L, int C, int N, int taps, int pad, float *out, void *dgrad_pack, int64_t dgrad_pack_bytes,
                           void *ws, int64_t ws_bytes, void *stream);
int sf_op_conv1d_bwd_cl_p(int dtype, const float *x, const float *act, const float *stats, const float *w, const void *dgrad_pack, const float *gamma,
                          const float *beta, int groups, float eps, const float *dy, const float *dx_add, int B, int L, int C, int N, int taps, int pad,
                          float *dx, float *dw, float *db, float *dgb, void *ws, int64_t ws_bytes, void *stream);
/* The whole weight set of a training step in ONE pack launch.  sf_op_conv1d_train_images: which images of `w` the convolution (forward, and
 * its data gradient) reads at this geometry -- bit 0 fw = [N][taps][C] fp32, bit 1 fwx = its split fp16 image, bit 2 dg = [C][taps][N] fp32
 * (taps flipped), bit 3 dgx = its split bf16 image; bit 4: not plannable (the launch wants the fragment-ordered image: use
 * sf_op_conv1d_train_fwd); < 0 on error.  `w` is read for its address alignment only (an aligned fp32 1x1 weight IS its own fw).
 * sf_train_pack_many: desc_dev = n_items x 7 64-bit words in DEVICE memory per weight -- the addresses w, fw, fwx, dg, dgx (0 = not
 * written), then N | C << 32, then taps | first_tile << 32 -- sorted by first_tile; a weight has ceil(C / 32) * ceil(N / 32) tiles,
 * total_tiles is their sum.  sf_op_conv1d_train_fwd_pk = sf_op_conv1d_train_fwd reading fw / fwx instead of packing; the backward pass
 * takes the weight's [dg | dgx] region (dgx at dg + 4 * C * taps * N bytes) as the dgrad_pack of sf_op_conv1d_bwd_cl_p. */
int sf_op_conv1d_train_images(int dtype, const float *w, int B, int L, int C, int N, int taps, int pad, int groups);
int sf_train_pack_many(const void *desc_dev, int n_items, int total_tiles, void *stream);
int sf_op_conv1d_train_fwd_pk(int dtype, const float *x, const float *w, const float *fw, const void *fwx, const float *bias, const float *gamma,
                              const float *beta, int groups, float eps, const float *residual, int B, int L, int C, int N, int taps, int pad, float *out,
                              void *ws, int64_t ws_bytes, void *stream);
/* dx_add (or NULL), in sf_op_conv1d_bwd_cl_p (GroupNorm convolutions only) and sf_op_ln_modulate_bwd_add: a second gradient of x -- the one
 * arriving through the residual connection that bypasses the op (ResnetItem: x + conv2(...conv1(x)); attention: x + to_out(attn(LN(x)))) --
 * is added to dx inside the normalisation backward's own pass over the tensor, instead of by a separate element-wise launch. */
int sf_op_ln_modulate_bwd_add(const float *x, const float *scale_shift, const float *dy, const float *dx_add, float eps, int B, int L, int C, float *dx,
                              float *dss, void *ws, int64_t ws_bytes, void *stream);
/* Length reductions of the training composition (fp32, channels-last): out[b][c] = sum_l x[b][l][c] * (y ? y[b][l][c] : 1) -- the
 * gradient of a per-clip broadcast add (cross-attention over one context token) and of the SkipModulate scale
 * (a-unet SkipModulate: x + scale[:, None, :] * h).  Two deterministic stages, no atomics.
 * Any C >= 1 (16-byte vector passes where C / 4 divides 256, one column per lane otherwise).  ws >= sf_op_length_sums_workspace_bytes. */
int64_t sf_op_length_sums_workspace_bytes(int B, int L, int C);
int sf_op_length_sums(const float *x, const float *y /* or NULL */, int B, int L, int C, float *out /* (B, C) */, void *ws, int64_t ws_bytes,
                      void *stream);
/* backward of sf_op_ln_modulate (fp32): dx:(B,L,C); dss:(B,2C) = [dscale | dshift] or NULL; ws >= sf_op_ln_modulate_bwd_workspace_bytes */
int64_t sf_op_ln_modulate_bwd_workspace_bytes(int B, int L, int C);
int sf_op_ln_modulate_bwd(const float *x, const float *scale_shift, const float *dy, float eps, int B, int L, int C, float *dx, float *dss,
                          void *ws, int64_t ws_bytes, void *stream);
/* backward of sf_op_attention (fp32 matrix cores, head_dim 64): out = the forward result; dq:(B,L,H*D), dkv:(B,L,2*H*D);
 * ws >= 2 * B * H * L floats (log-sum-exp and dO.O per query) */
int sf_op_attention_bwd(const float *q, const float *kv, const float *out, const float *dout, int B, int L, int heads, int head_dim, float *dq,
                        float *dkv, void *ws, int64_t ws_bytes, void *stream);
/* The same pair with the log-sum-exp of the scaled scores, lse:(B, heads, L), kept by the forward pass and handed to the backward pass
 * (one score pass less there); fp32, head_dim 64; ws >= B * heads * L * 4 bytes. */
int sf_op_attention_fwd_lse(const float *q, const float *kv, int B, int L, int heads, int head_dim, float *out, float *lse, void *stream);
/* the same with the arithmetic chosen by `dtype` (SF_F32, or SF_F32X: products from split fp16 operands) */
int sf_op_attention_fwd_lse_x(int dtype, const float *q, const float *kv, int B, int L, int heads, int head_dim, float *out, float *lse, void *stream);
int sf_op_attention_bwd_lse(const float *q, const float *kv, const float *out, const float *dout, const float *lse, int B, int L, int heads, int head_dim,
                            float *dq, float *dkv, void *ws, int64_t ws_bytes, void *stream);
/* the same with the arithmetic chosen by `dtype` (SF_F32, or SF_F32X: scores from split fp16 operands, the gradient products from split bf16 operands) */
int sf_op_attention_bwd_lse_x(int dtype, const float *q, const float *kv, const float *out, const float *dout, const float *lse, int B, int L, int heads,
                              int head_dim, float *dq, float *dkv, void *ws, int64_t ws_bytes, void *stream);
/* Kernel tuning aid: average milliseconds of `iters` back-to-back launches of one channels-last conv1d
 * (x:(B,L,C) -> (B,L*upsample,N), `taps` taps, bias + residual epilogue) with a forced kernel family
 * (path 0 auto, 1 classic, 2 wave-split-K, 4 v2), tile variant (-1 auto) and grid split-K factor (-1 auto). */
int sf_bench_conv1d(int dtype, int B, int L, int C, int N, int taps, int upsample, int path, int tile, int sk, int iters,
                    float *ms_out);
/* The deep-level item head as the small-batch engine runs it (conv_cb.hip; a-unet ResnetItem + ModulationItem, SURVEY appendix A.3
 * items 1-2), 16-bit dtypes, C a multiple of 128, L >= 44:
 *   h = conv3(silu(groupnorm(x; gn1)), w1) + b1;  y = x + conv3(silu(groupnorm(h; gn2)), w2) + b2;
 *   m = layer_norm(y; eps_ln, no affine) * (1 + scale[b]) + shift[b]          (scale_shift (B, 2C) or NULL -> plain normalise)
 * as: gn_silu -> channel-block split-K convolution (fp32 partial slabs) -> slab reduction + bias + GroupNorm chunk sums -> the same
 * convolution with the GroupNorm+SiLU panel prologue -> slab reduction + bias + residual + LayerNorm + Modulation.
 * x, h_out (optional), m_out: (B, L, C) channels-last in `dtype`; w1, w2: (C, C, 3) fp32 PyTorch layout. */
int64_t sf_op_resnet_mod_cb_workspace_bytes(int B, int L, int C);
int sf_op_resnet_mod_cb(int dtype, const void *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *gn1_g,
                        const float *gn1_b, const float *gn2_g, const float *gn2_b, int groups, float eps_gn, const float *scale_shift,
                        float eps_ln, int B, int L, int C, int kb /* 128-channel blocks per workgroup: 1 or 2 */, void *h_out, void *m_out,
                        void *ws, int64_t ws_bytes, void *stream);
/* InjectChannels followed by the attention pre-norm projection, as the engine chains the two GEMMs of an item (a-unet InjectChannelsItem
 * + the LayerNorm / to_q | to_kv Linear of AttentionItem), 16-bit dtypes:
 *   z = m + Conv1x1(cat[m, ctx]) + b_inj            m:(B,L,C), ctx:(B,L,C2) channels-last in `dtype`; w_inj:(C, C + C2) fp32
 *   q = Linear(LayerNorm_C(z; gamma, beta, eps))    w_q:(N, C) fp32, bias-free
 * The first GEMM's epilogue leaves per-row LayerNorm partials per 32-column tile, the second multiplies the RAW rows of z and
 * normalises its accumulator, rstd * (acc - mean * colsum) -- no LayerNorm launch in between.  Which kernel family runs (macro tiles at
 * long activations, the 32x32 families at short ones) follows the engine's dispatch, except that the macro-tile form is always offered
 * here (the engine does not take it: in the two-branch step it measured no gain); SF_ERR_UNSUPPORTED where no kernel fits.
 * fused_out (optional): set to 1 when the fused pair ran, 0 when the op fell back to z -> ln_modulate -> plain projection. */
int64_t sf_op_inject_prenorm_proj_workspace_bytes(int B, int L, int C, int C2, int N);
int sf_op_inject_prenorm_proj(int dtype, const void *m, const void *ctx, const float *w_inj, const float *b_inj, const float *gamma,
                              const float *beta, float eps, const float *w_q, int B, int L, int C, int C2, int N, void *z_out, void *q_out,
                              int *fused_out, void *ws, int64_t ws_bytes, void *stream);
/* Kernel tuning aid: average milliseconds of the four launches of that chain on random data (ms[0] convolution, ms[1] reduction +
 * GroupNorm sums, ms[2] convolution with prologue, ms[3] reduction + LayerNorm); cold != 0 streams the weights from HBM. */
int sf_bench_conv_cb(int dtype, int B, int L, int C, int groups, int kb, int cold, int iters, float *ms /* [4] */);
/* y = layer_norm(x; eps, no affine) * (1 + scale[b]) + shift[b]   (scale/shift NULL -> plain normalise) */
/* materialised SiLU(GroupNorm(x)) on channels-last rows (B, L, C), as the wide U-Net levels run it in front of their convolutions
 * (a-unet ResnetItem).  ws (optional, >= B * 32 * groups * 2 floats): long sequences then take the
 * chunked form (per-chunk statistics + one streaming pass) instead of one workgroup per (clip, group). */
int sf_op_gn_silu(int dtype, const void *x, const float *gamma, const float *beta, int groups, float eps, int B, int L, int C, void *out,
                  void *ws, int64_t ws_bytes, void *stream);
int sf_op_ln_modulate(int dtype, const void *x, const float *scale_shift /* (B, 2C) or NULL */, float eps,
                      int B, int L, int C, void *out, void *stream);
/* multi-head attention on packed projections: q:(B,L,H*D), kv:(B,L,2*H*D) -> out:(B,L,H*D) */
int sf_op_attention(int dtype, const void *q, const void *kv, int B, int L, int heads, int head_dim, void *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SYNCFUSION_AMD_H */
