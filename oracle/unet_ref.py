"""Oracle: fp32 CPU restatement of ``audio_diffusion_pytorch.UNetV0`` (a-unet XUNet).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED: the source
of ``a-unet`` / ``audio-diffusion-pytorch==0.1.3`` is not in /root/reference
(requirements.txt:23); this follows SURVEY.md appendix A.3 and the reference's
config ``exp/model/diffusion.yaml:11-33``.

Everything here is channels-first ``(B, C, L)`` exactly as the reference runs
it, written with ``torch.nn.functional`` on a flat ``{name: tensor}`` dict so
that it shares no code with the product's channels-last HIP engine.  The
cross-attention is evaluated *faithfully* (q/k/softmax over the single CLAP
token) -- the HIP engine collapses it to a bias; the parity tests prove the two
agree.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# exp/model/diffusion.yaml:16-33 (reference config) + a-unet defaults (A.3)
DEFAULT_CONFIG = dict(
    in_channels=1,
    channels=[8, 32, 64, 128, 256, 512, 1024, 1024],
    factors=[1, 4, 4, 4, 2, 2, 2, 2],
    items=[1, 2, 2, 2, 2, 2, 2, 4],
    attentions=[0, 0, 0, 0, 1, 1, 1, 1],
    cross_attentions=[1, 1, 1, 1, 1, 1, 1, 1],
    attention_heads=8,
    attention_features=64,
    context_channels=[2, 8, 16, 32, 64, 128, 256, 256],
    embedding_features=512,
    embedding_max_length=1,
    modulation_features=1024,
    resnet_groups=8,
)


# [RECALLED] facts about a-unet that neither shapes nor the reference's call sites can settle.  Each is a switch (cfg["variants"],
# a dict overriding these defaults) so that tools/pin_upstream.py can decide it NUMERICALLY against the installed upstream packages
# (SURVEY.md 8f-1), and tests/test_oracle_cpu.py checks that every switch changes the output (a wrong default cannot hide).
RECALLED_DEFAULTS = dict(
    skip_form="skip_plus_scaled_h",   # SkipModulate: skip + s*h (default) | "h_plus_scaled_skip": h + s*skip
    mod_ln_eps=1e-6,                  # ModulationItem LayerNorm eps: 1e-6 (default) | 1e-5 (torch default)
    mod_ln_affine=False,              # ModulationItem LayerNorm affine: off (default) | on (uses `<item>.mod.norm.weight/.bias` when present)
    mod_act="silu",                   # features -> SiLU -> Linear (default) | "none": Linear on the raw features
    attn_pos_embedding=False,         # Attention adds a learned positional embedding `<attn>.pos.weight` (max_length, C) to its input: off (default)
    attn_scale="head",                # logits * head_features**-0.5 (default) | "none"
    upsample_mode=None,               # None: cfg["upsample_mode"] (default "nearest" = UpsampleInterpolate) | "transpose" (Upsample)
    time_first_act=None,              # None: cfg["time_first_activation"] (default True = GELU after the time embedder's Linear) | "gelu" | "none"
)
# Two more [RECALLED] facts are decided by the PARAMETERS handed in, not by a switch (a checkpoint's shapes settle them, keymap.infer_variants):
# the number of learned Fourier frequencies of the time embedder (`net.time.fourier_w`: modulation_features // 2 = 512 per SURVEY appendix A,
# 128 if upstream's NumberEmbedder(dim=256) is what builds it) and a bias on every attention `to_out` Linear (`<attn>.to_out.bias` present).


def recalled_variants(cfg) -> Dict:
    v = dict(RECALLED_DEFAULTS)
    v.update(cfg.get("variants") or {})
    unknown = set(v) - set(RECALLED_DEFAULTS)
    assert not unknown, f"unknown oracle variant switches: {sorted(unknown)}"
    return v


def _lin(P: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def time_features(P: Dict[str, Tensor], sigma: Tensor, first_act: bool = True) -> Tensor:
    """TimeConditioningPlugin (A.3): learned-Fourier(sigma) -> Linear -> [GELU] -> 2x(Linear, GELU); the number of frequencies is whatever
    `net.time.fourier_w` holds."""
    w = P["net.time.fourier_w"]
    x = sigma.reshape(-1, 1).to(torch.float32)
    freqs = x * w[None, :] * (2.0 * math.pi)
    four = torch.cat([x, freqs.sin(), freqs.cos()], dim=-1)
    f = _lin(P, "net.time.lin0", four)
    if first_act:
        f = F.gelu(f)
    for i in range(2):
        f = F.gelu(_lin(P, f"net.time.mlp.{i}", f))
    return f


def _resnet(P, pre: str, x: Tensor, groups: int) -> Tensor:
    # ResnetItem (A.3 item 1): x + Conv3(SiLU(GN(Conv3(SiLU(GN(x))))))
    h = F.group_norm(x, groups, P[pre + ".gn1.weight"], P[pre + ".gn1.bias"], eps=1e-5)
    h = F.conv1d(F.silu(h), P[pre + ".conv1.weight"], P[pre + ".conv1.bias"], padding=1)
    h = F.group_norm(h, groups, P[pre + ".gn2.weight"], P[pre + ".gn2.bias"], eps=1e-5)
    h = F.conv1d(F.silu(h), P[pre + ".conv2.weight"], P[pre + ".conv2.bias"], padding=1)
    return x + h


def _modulation(P, pre: str, x: Tensor, f: Tensor, var=RECALLED_DEFAULTS) -> Tensor:
    # ModulationItem (A.3 item 2): LN_C(x; eps 1e-6, no affine) * (1 + s) + t
    C = x.shape[1]
    ss = _lin(P, pre + ".to_scale_shift", F.silu(f) if var["mod_act"] == "silu" else f)
    scale, shift = ss[:, None, :].chunk(2, dim=-1)
    xt = x.transpose(1, 2)
    g, b = (P.get(pre + ".norm.weight"), P.get(pre + ".norm.bias")) if var["mod_ln_affine"] else (None, None)
    xt = F.layer_norm(xt, (C,), g, b, eps=var["mod_ln_eps"]) * (1.0 + scale) + shift
    return xt.transpose(1, 2)


def _inject(P, pre: str, x: Tensor, ctx: Tensor) -> Tensor:
    # InjectChannelsItem (A.3 item 3): Conv1x1(cat[x, ctx]) + x
    assert ctx.shape[0] == x.shape[0] and ctx.shape[2] == x.shape[2], (ctx.shape, x.shape)
    return F.conv1d(torch.cat([x, ctx], dim=1), P[pre + ".conv.weight"], P[pre + ".conv.bias"]) + x


def _attention(P, pre: str, x: Tensor, context: Optional[Tensor], heads: int, head_features: int, var=RECALLED_DEFAULTS) -> Tensor:
    # AttentionItem / CrossAttentionItem (A.3 items 4-5), channels-last internally.
    xt = x.transpose(1, 2)  # (B, L, C)
    C = xt.shape[-1]
    if var["attn_pos_embedding"] and (pre + ".pos.weight") in P:
        xt = xt + P[pre + ".pos.weight"][: xt.shape[1]][None]
    ctx = xt if context is None else context
    q_in = F.layer_norm(xt, (C,), P[pre + ".norm.weight"], P[pre + ".norm.bias"], eps=1e-5)
    c_in = F.layer_norm(ctx, (ctx.shape[-1],), P[pre + ".norm_context.weight"], P[pre + ".norm_context.bias"], eps=1e-5)
    q = F.linear(q_in, P[pre + ".to_q.weight"])
    k, v = F.linear(c_in, P[pre + ".to_kv.weight"]).chunk(2, dim=-1)
    B, n, _ = q.shape
    m = k.shape[1]
    q = q.reshape(B, n, heads, head_features).transpose(1, 2)
    k = k.reshape(B, m, heads, head_features).transpose(1, 2)
    v = v.reshape(B, m, heads, head_features).transpose(1, 2)
    sim = torch.einsum("bhnd,bhmd->bhnm", q, k) * ((head_features ** -0.5) if var["attn_scale"] == "head" else 1.0)
    attn = sim.softmax(dim=-1, dtype=torch.float32)
    out = torch.einsum("bhnm,bhmd->bhnd", attn, v)
    out = out.transpose(1, 2).reshape(B, n, heads * head_features)
    out = F.linear(out, P[pre + ".to_out.weight"], P.get(pre + ".to_out.bias"))
    return (xt + out).transpose(1, 2)


def _item_group(P, cfg, pre: str, d: int, x, f, emb, ctx):
    var = recalled_variants(cfg)
    x = _resnet(P, pre + ".resnet", x, cfg["resnet_groups"])
    x = _modulation(P, pre + ".mod", x, f, var)
    if cfg["context_channels"][d] > 0:
        x = _inject(P, pre + ".inject", x, ctx[d])
    if cfg["attentions"][d]:
        x = _attention(P, pre + ".attn", x, None, cfg["attention_heads"], cfg["attention_features"], var)
    if cfg["cross_attentions"][d]:
        x = _attention(P, pre + ".cross", x, emb, cfg["attention_heads"], cfg["attention_features"], var)
    return x


def _block(P, cfg, d: int, x, f, emb, ctx, taps=None):
    pre = f"net.blocks.{d}"
    fac = cfg["factors"][d]
    skip = x
    h = F.conv1d(x, P[pre + ".down.weight"], P[pre + ".down.bias"], stride=fac)
    if taps is not None:
        taps[f"d{d}.down"] = h
    for j in range(cfg["items"][d]):
        h = _item_group(P, cfg, f"{pre}.items_down.{j}", d, h, f, emb, ctx)
        if taps is not None:
            taps[f"d{d}.items_down.{j}"] = h
    if d + 1 < len(cfg["channels"]):
        h = _block(P, cfg, d + 1, h, f, emb, ctx, taps)
    for j in range(cfg["items"][d]):
        h = _item_group(P, cfg, f"{pre}.items_up.{j}", d, h, f, emb, ctx)
        if taps is not None:
            taps[f"d{d}.items_up.{j}"] = h
    var = recalled_variants(cfg)
    if (var["upsample_mode"] or cfg.get("upsample_mode", "nearest")) == "transpose":
        # a-unet `Upsample` (apex.py): ConvTranspose1d(C -> in, kernel_size=factor, stride=factor); weight (C, in, factor)
        h = F.conv_transpose1d(h, P[pre + ".up.weight"], P[pre + ".up.bias"], stride=fac)
    else:
        # a-unet `UpsampleInterpolate`: nn.Upsample(scale_factor, mode="nearest") then Conv1d(C -> in, 3, padding=1)
        h = F.interpolate(h, scale_factor=fac, mode="nearest")
        h = F.conv1d(h, P[pre + ".up.weight"], P[pre + ".up.bias"], padding=1)
    # SkipModulate: skip + Linear(SiLU(f))[:, :, None] * h
    scale = _lin(P, pre + ".skip.to_scale", F.silu(f) if var["mod_act"] == "silu" else f)
    out = (skip + scale[:, :, None] * h) if var["skip_form"] == "skip_plus_scaled_h" else (h + scale[:, :, None] * skip)
    if taps is not None:
        taps[f"d{d}.out"] = out
    return out


def xunet_forward(P, cfg, x: Tensor, features: Tensor, embedding: Tensor, channels: Sequence[Tensor], taps=None) -> Tensor:
    """XUNet.forward(x, features=, embedding=, channels=) (A.3)."""
    for d, c in enumerate(channels):
        want = (x.shape[0], cfg["context_channels"][d])
        assert tuple(c.shape[:2]) == want, f"context channels at depth {d}: {tuple(c.shape)} vs {want}"
    return _block(P, cfg, 0, x, features, embedding, list(channels), taps)


def unet_forward(P, cfg, x: Tensor, sigma: Tensor, *, embedding: Tensor, channels: Sequence[Tensor],
                 embedding_scale: float = 1.0, taps=None) -> Tensor:
    """TimeConditioningPlugin(ClassifierFreeGuidancePlugin(XUNet)) forward (A.3).

    Two *sequential* passes when embedding_scale != 1, like upstream."""
    assert embedding is not None, "ClassifierFreeGuidancePlugin requires embedding"
    tfa = recalled_variants(cfg)["time_first_act"]
    f = time_features(P, sigma, cfg.get("time_first_activation", True) if tfa is None else tfa == "gelu")
    if embedding_scale != 1.0:
        B, n = embedding.shape[:2]
        fixed = P["net.cfg.fixed_embedding.weight"][:n][None].expand(B, -1, -1)
        out = xunet_forward(P, cfg, x, f, embedding, channels, taps)
        out_masked = xunet_forward(P, cfg, x, f, fixed, channels)
        return out_masked + (out - out_masked) * embedding_scale
    return xunet_forward(P, cfg, x, f, embedding, channels, taps)


# --------------------------------------------------------------------------
# analytic work model -- used by bench.py for rooflines
# --------------------------------------------------------------------------
def unet_flops_per_eval(cfg, L0: int) -> float:
    """ALGORITHMIC FLOPs of one U-Net evaluation of one clip (cross-attn collapsed)."""
    ch, fac, items = cfg["channels"], cfg["factors"], cfg["items"]
    ctx, att = cfg["context_channels"], cfg["attentions"]
    hd = cfg["attention_heads"] * cfg["attention_features"]
    total = 0.0
    L = L0
    cin = cfg["in_channels"]
    Lprev = L0
    for d in range(len(ch)):
        C = ch[d]
        L = Lprev // fac[d]
        g = 2 * items[d]
        total += 2.0 * L * C * cin * fac[d]                      # down conv
        total += g * (12.0 * L * C * C + 2.0 * L * (C + ctx[d]) * C + L * C)
        if att[d]:
            total += g * (2.0 * L * C * 3 * hd + 2.0 * L * hd * C + 4.0 * L * L * hd)
        total += (2.0 if cfg.get("upsample_mode", "nearest") == "transpose" else 6.0) * Lprev * cin * C   # up conv at the outer length
        cin = C
        Lprev = L
    return total
