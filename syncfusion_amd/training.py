"""The training step of the reference: a differentiable fp32 forward of the U-Net and of the onset
encoder whose heavy operations run -- forward AND backward -- in the HIP library.

What the reference trains (main/module_diffusion.py:73-82, exp/train_diffusion_gh.yaml:87 ``precision: 32``):

    _, info = onsets_encoder(y, with_info=True)                              # Encoder1d, TRAINED (configure_optimizers :53-61)
    loss = DiffusionModel(x, channels=info["xs"][2:-1], embedding=clap(z))   # VDiffusion: mse(net(alpha x + beta eps, sigma), v)

The inference engine behind ``UNetV0.forward`` / ``Encoder1d.forward`` is a fused, graph-replayed pipeline without an autograd
graph.  Here the same networks are composed per operation from ``syncfusion_amd.autograd`` (convolutions incl. the fused
GroupNorm + SiLU prologue, LayerNorm-modulate, multi-head attention: ``torch.autograd.Function``s over the C ABI, fp32, no
atomics so gradients are reproducible bit for bit) on channels-last ``(B, L, C)`` activations.  torch itself is used for what
the brief calls plumbing: the (B x features) conditioning Linears of the time MLP / modulation / skip scales (library GEMMs with
M = batch), reshapes / gathers around the patchify, strided and up-sampling convolutions, and the residual / skip additions.

The module's ``nn.Parameter`` masters are used directly, so ``loss.backward()`` fills ``.grad`` of exactly the tensors
``Model.configure_optimizers`` hands to AdamW.  ``UNetV0.forward`` / ``Encoder1d.forward`` route here whenever autograd is
recording and one of their parameters or inputs requires a gradient; otherwise the inference engine runs.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from . import autograd as sfa

Tensor = torch.Tensor


_warned_outside_step = False
_in_training_step = 0


class training_step_scope:
    """Marks the calls made from ``Model.training_step`` / ``fit_batches``: there the differentiable composition is what the
    caller asked for.  Anywhere else a forward call that records a graph gets ONE warning (see ``wants_grad``)."""

    def __enter__(self):
        global _in_training_step
        _in_training_step += 1
        return self

    def __exit__(self, *exc):
        global _in_training_step
        _in_training_step -= 1


def wants_grad(module: torch.nn.Module, *tensors: Optional[Tensor]) -> bool:
    """True when a forward call of ``module`` has to record an autograd graph.

    A freshly built module has trainable parameters and sits in train() mode, so ANY call outside ``torch.no_grad()`` lands
    here -- and then leaves the 16-bit graph engine for the fp32 per-operation composition that keeps every activation (GBs at
    2^18 samples).  That is right inside a training step and almost always an accident elsewhere, so the first such call made
    outside ``Model.training_step`` / ``fit_batches`` warns (eval() or train() mode alike)."""
    global _warned_outside_step
    if not torch.is_grad_enabled():
        return False
    want = any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())
    if want and not _in_training_step and not _warned_outside_step:
        _warned_outside_step = True
        import warnings

        warnings.warn(f"{type(module).__name__}: autograd is recording outside Model.training_step, so this call runs the differentiable fp32 "
                      "composition (syncfusion_amd.training: every activation is kept for backward), not the inference engine.  Wrap "
                      "inference calls in torch.no_grad().", stacklevel=3)
    return want


def _params(module: torch.nn.Module) -> Dict[str, Tensor]:
    return dict(module.named_parameters())


def _lin(P, name: str, x: Tensor) -> Tensor:
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _pointwise(x: Tensor, w: Tensor, b: Optional[Tensor], residual: Optional[Tensor] = None) -> Tensor:
    """Linear / 1x1 convolution over the channels of (B, L, C) rows (+ residual, added in the kernel's epilogue): w is (N, C) or (N, C, 1)."""
    return sfa.conv1d(x, w.reshape(w.shape[0], -1, 1), b, channels_last=True, residual=residual)


def _affine_ln(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, passthrough: bool = False):
    """LayerNorm_C with affine parameters, as LN-modulate with scale = gamma - 1, shift = beta for every clip."""
    # every clip shares (gamma, beta): the (B, L, C) rows are ONE clip of B * L rows to the kernel -- no per-clip copy of the pair in forward,
    # no sum over the clips in backward (the kernel's length reduction covers all rows)
    B, L, C = x.shape
    ss = torch.cat([gamma - 1.0, beta])[None, :]
    if passthrough:
        y, xp = sfa.ln_modulate(x.reshape(1, B * L, C), ss, eps, passthrough=True)
        return y.reshape(B, L, C), xp.reshape(B, L, C)
    return sfa.ln_modulate(x.reshape(1, B * L, C), ss, eps).reshape(B, L, C)


# ---------------------------------------------------------------------------------------------------------------------------
# U-Net (a-unet XUNet behind TimeConditioningPlugin + ClassifierFreeGuidancePlugin)
# ---------------------------------------------------------------------------------------------------------------------------
def _time_features(P, sigma: Tensor, first_act: bool = True) -> Tensor:
    w = P["time.fourier_w"]
    x = sigma.reshape(-1, 1).to(torch.float32)
    freqs = x * w[None, :] * (2.0 * math.pi)
    f = _lin(P, "time.lin0", torch.cat([x, freqs.sin(), freqs.cos()], dim=-1))
    if first_act:                       # [RECALLED] switch hparams["time_first_activation"] (UNetV0)
        f = F.gelu(f)
    for i in range(2):
        f = F.gelu(_lin(P, f"time.mlp.{i}", f))
    return f


def _resnet(P, pre: str, x: Tensor, groups: int) -> Tensor:
    # (x reaches the residual connection THROUGH conv1's node: both of its gradients meet in that node's GroupNorm backward kernel)
    h, x = sfa.gn_silu_conv1d(x, P[pre + ".conv1.weight"], P[pre + ".conv1.bias"], P[pre + ".gn1.weight"], P[pre + ".gn1.bias"], groups, 1e-5, True,
                              passthrough=True)
    return sfa.gn_silu_conv1d(h, P[pre + ".conv2.weight"], P[pre + ".conv2.bias"], P[pre + ".gn2.weight"], P[pre + ".gn2.bias"], groups, 1e-5, True,
                              residual=x)


def _self_attention(P, pre: str, x: Tensor, heads: int) -> Tensor:
    # (x is handed from node to node -- q norm, then k/v norm, then the residual: its three gradients are summed inside the two LayerNorm
    #  backward kernels instead of by two element-wise launches of the autograd engine)
    xq, x = _affine_ln(x, P[pre + ".norm.weight"], P[pre + ".norm.bias"], 1e-5, passthrough=True)
    xkv, x = _affine_ln(x, P[pre + ".norm_context.weight"], P[pre + ".norm_context.bias"], 1e-5, passthrough=True)
    q = _pointwise(xq, P[pre + ".to_q.weight"], None)
    kv = _pointwise(xkv, P[pre + ".to_kv.weight"], None)
    return _pointwise(sfa.attention(q, kv, heads), P[pre + ".to_out.weight"], P.get(pre + ".to_out.bias"), residual=x)


class _ZeroGradAnchor(torch.autograd.Function):
    """``y = x``; the listed parameters join the graph and receive exactly-zero gradients in backward (one fill for all) -- what upstream's
    autograd gives parameters whose branch cannot influence the output.  (Anchoring them with ``0 * p.sum()`` terms cost a reduction per
    parameter in forward and an expand + multiply + accumulate in backward: 4 % of the training step in ATen reduce / fill kernels.)"""

    @staticmethod
    def forward(ctx, x, *params):
        ctx.meta = [(p.shape, p.dtype, p.device) for p in params]
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # ONE fill for all of them: the gradients are views of a single zero buffer (per dtype / device)
        pools, outs = {}, []
        for shape, dt, dev in ctx.meta:
            pools[(dt, dev)] = pools.get((dt, dev), 0) + (math.prod(shape) + 3) // 4 * 4
        bufs = {k: [torch.zeros(n, dtype=k[0], device=k[1]), 0] for k, n in pools.items()}
        for shape, dt, dev in ctx.meta:
            buf = bufs[(dt, dev)]
            n = math.prod(shape)
            outs.append(buf[0][buf[1]:buf[1] + n].view(shape))
            buf[1] += (n + 3) // 4 * 4
        return (g,) + tuple(outs)


_ATEN_SUMS = os.environ.get("SF_TRAIN_ATEN_SUMS") == "1"   # A/B aid: the length reductions through ATen as before


class _ClipAdd(torch.autograd.Function):
    """``x + o`` with o (B, 1, C) broadcast over the length; backward sums over the length in one HIP pass (``sfa.length_sums``: ATen's
    strided reduction over the middle dimension ran at ~0.3 TB/s on the wide levels, 2.2 ms of a 76 ms step)."""

    @staticmethod
    def forward(ctx, x, o):
        return x + o

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[1]:
            return g, None
        return g, (g.sum(dim=1, keepdim=True) if _ATEN_SUMS else sfa.length_sums(g)[:, None, :])


class _SkipModulate(torch.autograd.Function):
    """``x + scale[:, None, :] * h`` (a-unet SkipModulate) as one fused multiply-add; the scale gradient sum_l g * h in one HIP pass."""

    @staticmethod
    def forward(ctx, x, scale, h):
        ctx.save_for_backward(scale, h)
        return torch.addcmul(x, scale[:, None, :], h)

    @staticmethod
    def backward(ctx, g):
        scale, h = ctx.saved_tensors
        gs = ((g * h).sum(dim=1) if _ATEN_SUMS else sfa.length_sums(g, h)) if ctx.needs_input_grad[1] else None
        gh = g * scale[:, None, :] if ctx.needs_input_grad[2] else None
        return g, gs, gh


_PACK_PLANS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_ITEM_MAJOR: Dict[tuple, Tensor] = {}


def _pack_plan(module, x: Tensor) -> "sfa.PackPlan":
    """every convolution weight of a pass through ``module`` packed by ONE launch at its start (``autograd.PackPlan``: recorded on the first
    pass of an input shape; weakly keyed by the module -- the plans' buffers are no part of its state, copies or pickles)"""
    plans = _PACK_PLANS.setdefault(module, {})
    sig = (tuple(x.shape), str(x.device), sfa.GEMM_DTYPE, torch.is_grad_enabled())
    if sig not in plans:
        if len(plans) >= 4:
            plans.clear()
        plans[sig] = sfa.PackPlan()
    return plans[sig]


def _item_major_index(B: int, sizes: tuple, device) -> Tensor:
    """flat position of element (item i, clip b, column k) in a row-major (B, sum(sizes)) matrix, items outermost (cached per shape)"""
    key = (B, sizes, str(device))
    idx = _ITEM_MAJOR.get(key)
    if idx is None:
        total, off, parts = sum(sizes), 0, []
        for sz in sizes:
            parts.append((torch.arange(B)[:, None] * total + off + torch.arange(sz)[None, :]).reshape(-1))
            off += sz
        idx = _ITEM_MAJOR[key] = torch.cat(parts).to(device)
    return idx


def _modulation_outputs(P, hp, f_act: Tensor) -> Dict[str, Tensor]:
    """Every Modulation (to_scale_shift) and SkipModulate (to_scale) Linear of the net reads the same SiLU(time features): one Linear over
    the concatenated weights (the inference engine keeps them as one (34569 x 1024) matrix too), split into per-item views -- 42 small
    addmm forward and their 3 backward launches each become one of each; the concatenation itself is one 141 MB copy per step."""
    names = []   # the (2C)-wide Modulation rows first (multiples of 16 floats: every view stays 16-byte aligned), the SkipModulate rows last
    for d in range(len(hp["channels"])):
        for part in ("items_down", "items_up"):
            names += [f"blocks.{d}.{part}.{j}.mod.to_scale_shift" for j in range(hp["items"][d])]
    names += [f"blocks.{d}.skip.to_scale" for d in range(len(hp["channels"]))]
    w = torch.cat([P[n + ".weight"] for n in names])
    b = torch.cat([P[n + ".bias"] for n in names])
    out = F.linear(f_act, w, b)
    sizes = [int(P[n + ".weight"].shape[0]) for n in names]
    # the consumers read contiguous (B, width) rows: repack the column slices item-major with ONE gather (a permutation of out's elements;
    # its backward is one scatter of unique indices: deterministic) instead of one strided copy per consumer and direction
    B = out.shape[0]
    flat = out.reshape(-1).index_select(0, _item_major_index(B, tuple(sizes), out.device))
    return {n: v.reshape(B, sz) for n, v, sz in zip(names, flat.split([B * sz for sz in sizes]), sizes)}


def _cross_attention_outputs(P, hp, emb: Tensor) -> Dict[str, Tensor]:
    """Cross-attention over ONE context token (embedding_max_length = 1, exp/model/diffusion.yaml:30) for EVERY item of the net at once: the
    softmax over a single key is identically 1, so an item adds o = to_out(v(LN(emb))) (B, 1, C) to every position.  All items read the same
    embedding, so the 34 LayerNorms share one normalisation (their affines are applied as a stacked multiply-add), the value projections are
    one batched product and the output projections one batched product per channel count -- 21 small launches instead of 102 forward, and as
    many fewer backward (1.3 ms of launch-bound ATen kernels per training step).  The key half of to_kv receives exactly-zero gradients
    through the slice below, as upstream's autograd gives it."""
    pres = []
    for d in range(len(hp["channels"])):
        if hp["cross_attentions"][d]:
            for part in ("items_down", "items_up"):
                pres += [f"blocks.{d}.{part}.{j}.cross" for j in range(hp["items"][d])]
    if not pres:
        return {}
    if emb.shape[1] != 1:
        raise NotImplementedError("training forward: cross-attention over more than one embedding token is not implemented")
    hd = hp["attention_heads"] * hp["attention_features"]
    xhat = F.layer_norm(emb[:, 0, :], (emb.shape[-1],), None, None, eps=1e-5)                          # (B, E), shared by every item
    gam = torch.stack([P[p + ".norm_context.weight"] for p in pres])                                  # (n, E)
    bet = torch.stack([P[p + ".norm_context.bias"] for p in pres])
    c_in = torch.addcmul(bet[:, None, :], xhat[None], gam[:, None, :])                                 # (n, B, E)
    wv = torch.stack([P[p + ".to_kv.weight"] for p in pres])[:, hd:, :]                                # (n, hd, E): the value half only (one slice
                                                                                                       # of the stack: its backward is one zero-fill + copy, not 34)
    v = torch.bmm(c_in, wv.transpose(1, 2))                                                            # (n, B, hd)
    out: Dict[str, Tensor] = {}
    by_c: Dict[int, List[int]] = {}
    for i, p in enumerate(pres):
        by_c.setdefault(int(P[p + ".to_out.weight"].shape[0]), []).append(i)
    for _c, idx in by_c.items():
        wo = torch.stack([P[pres[i] + ".to_out.weight"] for i in idx])                                 # (k, C, hd)
        vi = v[idx[0]:idx[-1] + 1] if idx == list(range(idx[0], idx[-1] + 1)) else v[idx]
        o = torch.bmm(vi, wo.transpose(1, 2))                                                          # (k, B, C)
        if (pres[idx[0]] + ".to_out.bias") in P:                                                       # [RECALLED] switch attention_out_bias
            o = o + torch.stack([P[pres[i] + ".to_out.bias"] for i in idx])[:, None, :]
        for j, i in enumerate(idx):
            out[pres[i]] = o[j][:, None, :]
    return out


def _cross_attention(P, pre: str, x: Tensor, o: Tensor) -> Tensor:
    """Adds the item's collapsed cross-attention output (``_cross_attention_outputs``).  The query branch (norm, to_q) cannot influence a
    one-token softmax: it receives exactly-zero gradients upstream too, here through ``_ZeroGradAnchor`` -- the optimizer sees zero gradients
    (and applies weight decay) rather than ``None``."""
    o = _ZeroGradAnchor.apply(o, P[pre + ".to_q.weight"], P[pre + ".norm.weight"], P[pre + ".norm.bias"])
    return _ClipAdd.apply(x, o)


def _item_group(P, hp, pre: str, d: int, x: Tensor, f_act: Dict[str, Tensor], ca: Dict[str, Tensor], ctx: List[Tensor]) -> Tensor:
    x = _resnet(P, pre + ".resnet", x, hp["resnet_groups"])
    x = sfa.ln_modulate(x, f_act[pre + ".mod.to_scale_shift"], 1e-6)
    if hp["context_channels"][d] > 0:
        x = _pointwise(torch.cat([x, ctx[d]], dim=-1), P[pre + ".inject.conv.weight"], P[pre + ".inject.conv.bias"], residual=x)
    if hp["attentions"][d]:
        x = _self_attention(P, pre + ".attn", x, hp["attention_heads"])
    if hp["cross_attentions"][d]:
        x = _cross_attention(P, pre + ".cross", x, ca[pre + ".cross"])
    return x


def _block(P, hp, d: int, x: Tensor, f_act: Dict[str, Tensor], ca: Dict[str, Tensor], ctx: List[Tensor]) -> Tensor:
    pre = f"blocks.{d}"
    fac = hp["factors"][d]
    B, L, cin = x.shape
    if L % fac:
        raise ValueError(f"length {L} at depth {d} is not divisible by the down-sampling factor {fac}")
    wd = P[pre + ".down.weight"]                                               # (C, cin, fac): kernel = stride = fac, a patchify
    C = wd.shape[0]
    h = _pointwise(x.reshape(B, L // fac, fac * cin), wd.permute(0, 2, 1).reshape(C, fac * cin), P[pre + ".down.bias"])
    for j in range(hp["items"][d]):
        h = _item_group(P, hp, f"{pre}.items_down.{j}", d, h, f_act, ca, ctx)
    if d + 1 < len(hp["channels"]):
        h = _block(P, hp, d + 1, h, f_act, ca, ctx)
    for j in range(hp["items"][d]):
        h = _item_group(P, hp, f"{pre}.items_up.{j}", d, h, f_act, ca, ctx)
    wu, bu = P[pre + ".up.weight"], P[pre + ".up.bias"]
    if hp.get("upsample_mode", "nearest") == "transpose":
        # ConvTranspose1d(kernel = stride = fac), weight (C, cin, fac): every position emits fac outputs -> a pointwise map to
        # fac * cin features that is unfolded along the length
        h = _pointwise(h, wu.permute(2, 1, 0).reshape(fac * cin, C), bu.repeat(fac)).reshape(B, L, cin)
    else:
        if fac > 1:
            h = h.repeat_interleave(fac, dim=1)
        h = sfa.conv1d(h, wu, bu, channels_last=True)
    scale = f_act[pre + ".skip.to_scale"]                                       # SkipModulate
    return _SkipModulate.apply(x, scale, h)


def unet_forward(net, x: Tensor, sigma: Tensor, *, embedding: Tensor, channels: Sequence[Tensor], embedding_scale: float = 1.0,
                 embedding_mask_proba: float = 0.0) -> Tensor:
    """Differentiable ``UNetV0.forward``: x (B, in_channels, L0), sigma (B,), embedding (B, 1, E), channels[d] (B, ctx_d, L_d)
    -> v (B, in_channels, L0)."""
    _lib.require_gpu_tensor(x, "syncfusion_amd.training.unet_forward")
    hp = net.hparams
    P = _params(net)
    if len(channels) != len(hp["channels"]):
        raise ValueError(f"channels: expected {len(hp['channels'])} context tensors, got {len(channels)}")
    B = x.shape[0]
    for d, c in enumerate(channels):
        want = (B, hp["context_channels"][d])
        assert tuple(c.shape[:2]) == want, f"context channels at depth {d}: {tuple(c.shape)} vs {want}"
    ctx = [c.to(torch.float32).transpose(1, 2) for c in channels]
    f_act = _modulation_outputs(P, hp, F.silu(_time_features(P, sigma, hp.get("time_first_activation", True))))   # per-item modulation rows, keyed by Linear name
    emb = embedding.to(torch.float32)
    fixed = P["cfg.fixed_embedding.weight"][: emb.shape[1]][None].expand(B, -1, -1)
    if embedding_mask_proba > 0.0:   # ClassifierFreeGuidancePlugin: per-clip replacement by the learned fixed embedding
        mask = torch.rand(B, 1, 1, device=x.device) < embedding_mask_proba
        emb = torch.where(mask, fixed, emb)
    x_cl = x.to(torch.float32).transpose(1, 2)
    with _pack_plan(net, x):
        out = _block(P, hp, 0, x_cl, f_act, _cross_attention_outputs(P, hp, emb), ctx)
        if embedding_scale != 1.0:
            out_masked = _block(P, hp, 0, x_cl, f_act, _cross_attention_outputs(P, hp, fixed), ctx)
            out = out_masked + (out - out_masked) * embedding_scale
    return out.transpose(1, 2)


# ---------------------------------------------------------------------------------------------------------------------------
# Encoder1d (audio_encoders_pytorch)
# ---------------------------------------------------------------------------------------------------------------------------
def _enc_resnet(P, pre: str, x: Tensor, groups: int) -> Tensor:
    h = x
    for blk in ("block1", "block2"):
        h = sfa.gn_silu_conv1d(h, P[f"{pre}.{blk}.conv.weight"], P[f"{pre}.{blk}.conv.bias"], P[f"{pre}.{blk}.gn.weight"], P[f"{pre}.{blk}.gn.bias"],
                               groups, 1e-5, True)
    if (pre + ".to_out.weight") in P:
        x = _pointwise(x, P[pre + ".to_out.weight"], P[pre + ".to_out.bias"])
    return h + x


def encoder1d_forward(enc, y: Tensor) -> Tuple[Tensor, Dict[str, List[Tensor]]]:
    """Differentiable ``Encoder1d.forward(y, with_info=True)``: (z, {"xs": [y, to_in, ds_0 .. ds_{n-1}, to_out]}), channels-first
    tensors exactly as the reference slices them (main/module_diffusion.py:76)."""
    _lib.require_gpu_tensor(y, "syncfusion_amd.training.encoder1d_forward")
    hp = enc.hparams
    P = _params(enc)
    x = y.to(torch.float32).transpose(1, 2)
    xs = [y]
    with _pack_plan(enc, y):
        x = _enc_resnet(P, "to_in", x, 1)
        xs.append(x.transpose(1, 2))
        for i, f in enumerate(hp["factors"]):
            pre = f"downsamples.{i}"
            w = P[pre + ".down.weight"]                                            # (N, C, 2f + 1), stride f, padding f
            N, Cc, k = w.shape
            cols = F.pad(x, (0, 0, f, f)).unfold(1, k, f)                           # (B, Lout, C, k) windows
            Bq, Lout = cols.shape[:2]
            cols = cols.permute(0, 1, 3, 2).reshape(Bq, Lout, k * Cc)
            x = _pointwise(cols, w.permute(0, 2, 1).reshape(N, k * Cc), P[pre + ".down.bias"])
            for j in range(hp["num_blocks"][i]):
                x = _enc_resnet(P, f"{pre}.blocks.{j}", x, hp["resnet_groups"])
            xs.append(x.transpose(1, 2))
    xs.append(xs[-1])
    return xs[-1], dict(xs=xs)


# ---------------------------------------------------------------------------------------------------------------------------
# data-parallel gradient exchange
# ---------------------------------------------------------------------------------------------------------------------------
def allreduce_gradients(module: torch.nn.Module, bucket_bytes: int = 256 << 20) -> int:
    """Average ``.grad`` over the ranks of the default process group in flat buckets (the reference trains with Lightning's
    single-node DDP defaults, exp/train_diffusion_gh.yaml:82-90).  One bucket of 256 MB amortises the per-link ring latency of
    xGMI; the whole model is 1.6 GB of fp32 gradients.  Returns the number of collectives issued (0 when not distributed)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    # EVERY trainable parameter takes part, a missing gradient as zeros INSIDE THE FLAT BUCKET only: the bucket boundaries (hence the
    # number and the sizes of the collectives) must not depend on which parameters happened to receive a gradient on THIS rank.
    # A presence flag per parameter rides with the first bucket: a parameter that no rank had a gradient for keeps `grad = None`
    # afterwards, exactly as on a single rank (the fused AdamW step skips it: no weight decay, no moment update).
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return 0
    dev = params[0].device
    present = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32, device=dev)
    stage_host = dist.get_backend() == "gloo"
    calls, i = 0, 0
    while i < len(params):
        j, size = i, 0
        while j < len(params) and (j == i or size + params[j].numel() * 4 <= bucket_bytes):
            size += params[j].numel() * 4
            j += 1
        parts = [(p.grad if p.grad is not None else torch.zeros(p.numel(), dtype=torch.float32, device=p.device)).reshape(-1).to(torch.float32)
                 for p in params[i:j]]
        if i == 0:
            parts.append(present)
        flat = torch.cat(parts)
        buf = flat.cpu() if stage_host and flat.is_cuda else flat
        dist.all_reduce(buf)
        buf = buf.to(flat.device)
        if i == 0:
            present = buf[-len(params):].cpu().tolist()   # one host read per call, not one per parameter
            buf = buf[:-len(params)]
        buf = buf / world
        off = 0
        for k, p in enumerate(params[i:j]):
            n = p.numel()
            if present[i + k] > 0.0:      # some rank had a gradient: the average (missing ranks count as zeros)
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(buf[off: off + n].view_as(p))
            off += n
        calls += 1
        i = j
    return calls


# ---------------------------------------------------------------------------------------------------------------------------
# the optimisation step as exp/train_diffusion_gh.yaml configures Lightning's Trainer for it
# ---------------------------------------------------------------------------------------------------------------------------
class GraphedTrainStep:
    """``Model.training_step`` + ``loss.backward()`` captured ONCE in a HIP graph and replayed per step (static batch shape).

    The training composition issues ~4400 launches per step from Python; with the fp32x GEMMs the GPU finishes a step in ~52 ms and
    the host needs ~60 ms to issue it, so the eager step is host-bound.  Replay removes the host from the step: one graph launch, then the
    optimizer.  What is captured is exactly ``Model.step`` (CLAP embedding of ``z``, onset encoder, v-objective loss on ``x``) and its
    backward; the per-step randomness of ``VDiffusion.forward`` (sigmas ~ U(0, 1), noise ~ N(0, 1): a-unet VDiffusion)
    lives in two static tensors that are refilled before every replay with torch's generator, so the step draws fresh values like the eager
    one.  Gradients land in the parameters' ``.grad`` tensors that the capture allocated (every replay overwrites them): call the optimizer
    (and ``allreduce_gradients`` / clipping) after ``step()`` as usual, and do NOT ``zero_grad(set_to_none=True)`` afterwards -- that would
    drop the tensors the graph writes.

    Build it BEFORE the first eager ``backward()`` of the process on these parameters (or after every reference to such a graph is gone):
    autograd's AccumulateGrad nodes remember the stream they were created on, and nodes left over from an eager backward on the default
    stream make the capture synchronise -- which HIP graph capture does not survive.

        gs = GraphedTrainStep(model, example_batch)       # captures; the batch tensors are copied into static buffers
        for batch in loader:
            loss = gs.step(batch)                         # copy-in, refill randomness, replay
            optimizer.step()
    """

    def __init__(self, model, batch, warmup: int = 2):
        x, y, z = batch[0], batch[1], batch[2]
        self.model = model
        self.x, self.y, self.z = x.clone(), y.clone(), (x if z is x else z).clone()
        if z is x:
            self.z = self.x
        self.sig = torch.rand(x.shape[0], device=x.device)
        self.noise = torch.randn_like(x)
        params = [p for p in model.parameters() if p.requires_grad]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), training_step_scope():   # warm-up on a side stream: allocator pools, lazy kernel loads, .grad allocation
            for _ in range(max(1, warmup)):
                for p in params:
                    p.grad = None
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), training_step_scope():
            self.loss = self._fwd_bwd()

    def _fwd_bwd(self) -> Tensor:
        m = self.model
        emb = m.clap_encode_audio(self.z)
        _, info = m.onsets_encoder(self.y, with_info=True)
        loss = m.model(self.x, channels=info["xs"][2:-1], embedding=emb, sigmas=self.sig, noise=self.noise)
        loss.backward()
        return loss

    def step(self, batch=None, resample: bool = True) -> Tensor:
        """Copy ``batch`` (same shapes as the example) into the static buffers, draw this step's sigmas / noise (``resample=False``: keep
        what ``self.sig`` / ``self.noise`` hold), replay.  Returns the static loss tensor (a device scalar: read it after the step, not
        inside a timed loop)."""
        if batch is not None:
            self.x.copy_(batch[0])
            self.y.copy_(batch[1])
            if self.z is not self.x:
                self.z.copy_(batch[2])
        if resample:
            self.sig.uniform_()
            self.noise.normal_()
        self.graph.replay()
        return self.loss


def fit_batches(model, optimizer, batches, *, accumulate_grad_batches: int = 2, gradient_clip_val: Optional[float] = 0.5,
                on_step=None) -> List[float]:
    """Run ``Model.training_step`` over ``batches`` the way the reference's trainer section does
    (exp/train_diffusion_gh.yaml:84-96: ``accumulate_grad_batches: 2``, ``gradient_clip_val: 0.5`` with Lightning's default
    clip-by-global-norm, fp32): the loss of every micro-batch is divided by the accumulation count, gradients are summed over
    ``accumulate_grad_batches`` micro-batches, averaged over the data-parallel ranks (``allreduce_gradients``), clipped to the
    global L2 norm and applied.  A trailing incomplete accumulation window is applied as Lightning does at the end of an epoch.
    Returns the micro-batch losses.  ``on_step(step_index, mean_loss)`` is called after every optimizer step."""
    if accumulate_grad_batches < 1:
        raise ValueError("accumulate_grad_batches must be >= 1")
    params = [p for group in optimizer.param_groups for p in group["params"]]
    losses: List[float] = []
    window: List[float] = []
    steps = 0

    def apply():
        nonlocal steps
        allreduce_gradients(model)
        if gradient_clip_val is not None and gradient_clip_val > 0:
            torch.nn.utils.clip_grad_norm_(params, gradient_clip_val)
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        steps += 1
        if on_step is not None:
            on_step(steps, sum(window) / len(window))
        window.clear()

    def everyone_has(have: bool) -> bool:
        """Ranks may see different batch counts (uneven shards): every rank takes a micro-batch only while ALL ranks still have
        one, so the number of gradient all-reduces is the same everywhere (Lightning's DDP join does the same job)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return have
        flag = torch.tensor([1 if have else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            flag = flag.to(params[0].device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag) == 1)

    optimizer.zero_grad(set_to_none=True)
    it = iter(batches)
    i = 0
    with training_step_scope():
        while True:
            try:
                batch = next(it)
                have = True
            except StopIteration:
                batch, have = None, False
            if not everyone_has(have):
                break
            loss = model.training_step(batch, i)
            (loss / accumulate_grad_batches).backward()
            window.append(float(loss.detach()))
            losses.append(window[-1])
            i += 1
            if len(window) == accumulate_grad_batches:
                apply()
        if window:
            apply()
    return losses
