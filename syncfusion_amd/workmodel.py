"""Closed-form ALGORITHMIC work of one U-Net evaluation, per depth.

Independent of how the engine splits the work (branches, tile shapes, fused or separate launches): FLOPs count
the contractions of the collapsed network (cross-attention over the single CLAP token = a bias; its Q / K /
out-projection FLOPs are NOT credited), bytes count every parameter ONCE per denoise step plus the fused-ideal
activation traffic.  ``bench.py`` prices a denoise step against the MI355X peaks with these numbers:

    t_roofline(step) = sum_d max( F_d / P_mfma , Q_d / BW_hbm )

Formulae (g_d = 2 * items_d item-groups, a_d = attentions_d, L_{-1} = L0, hd = heads * head_features):
    F_d = 2 L_d C_d Cin_d f_d                                     patchify down-conv
        + g_d (12 L_d C_d^2 + 2 L_d (C_d + ctx_d) C_d + L_d C_d)  two conv3, inject 1x1, modulation
        + a_d g_d (2 L_d C_d 3hd + 2 L_d hd C_d + 4 L_d^2 hd)     q|kv projection, out projection, QK^T + PV
        + 6 L_{d-1} Cin_d C_d                                     nearest-upsample + conv3 at the outer length
          (transposed-conv up path, kernel = stride = f:  2 L_{d-1} Cin_d C_d)
    Q_d = params_d * es                                           (weights: once per step, whatever the batch)
        + clips * evals * es * [ g_d ((6 + 2 a_d) L_d C_d + L_d ctx_d) + 3 L_{d-1} Cin_d + 2 L_d C_d ]
"""
from __future__ import annotations

from typing import Dict, List


def _params_by_depth(hp: Dict, upsample_mode: str = "nearest") -> List[float]:
    """Parameters stored under ``blocks.{d}.*`` (same layout as UNetV0._build), per depth."""
    mf, E = hp["modulation_features"], hp["embedding_features"]
    hd = hp["attention_heads"] * hp["attention_features"]
    out = []
    cin = hp["in_channels"]
    for d, C in enumerate(hp["channels"]):
        f = hp["factors"][d]
        n = C * cin * f + C                                   # down
        n += (cin * C * (f if upsample_mode == "transpose" else 3)) + cin   # up
        n += cin * mf + cin                                   # SkipModulate
        per = 2 * (2 * C) + 2 * (C * C * 3 + C)               # two GroupNorms, two conv3
        per += 2 * C * mf + 2 * C                             # Modulation
        ctx = hp["context_channels"][d]
        if ctx > 0:
            per += C * (C + ctx) + C                          # InjectChannels
        for feat, on in ((C, hp["attentions"][d]), (E, hp["cross_attentions"][d])):
            if on:
                per += 2 * C + 2 * feat + hd * C + 2 * hd * feat + C * hd
        n += 2 * hp["items"][d] * per
        out.append(float(n))
        cin = C
    return out


def unet_work(hp: Dict, L0: int, clips: int, evals: int, es: int, upsample_mode: str = "nearest") -> Dict:
    """Per-depth algorithmic FLOPs / bytes of ONE denoise step (``evals`` U-Net evaluations of ``clips`` clips)."""
    ch, fac, items = hp["channels"], hp["factors"], hp["items"]
    ctx, att = hp["context_channels"], hp["attentions"]
    hd = hp["attention_heads"] * hp["attention_features"]
    params = _params_by_depth(hp, upsample_mode)
    mf, E = hp["modulation_features"], hp["embedding_features"]
    glob = (mf // 2) + (mf * (mf + 1) + mf) + 2 * (mf * mf + mf) + hp["embedding_max_length"] * E
    n = float(clips * evals)
    flops, byts = [], []
    cin, Lprev = hp["in_channels"], L0
    for d, C in enumerate(ch):
        L = Lprev // fac[d]
        g = 2 * items[d]
        F = 2.0 * L * C * cin * fac[d]
        F += g * (12.0 * L * C * C + 2.0 * L * (C + ctx[d]) * C + L * C)
        if att[d]:
            F += g * (2.0 * L * C * 3 * hd + 2.0 * L * hd * C + 4.0 * L * L * hd)
        F += (2.0 if upsample_mode == "transpose" else 6.0) * Lprev * cin * C
        act = g * ((6 + 2 * (1 if att[d] else 0)) * L * C + L * ctx[d]) + 3.0 * Lprev * cin + 2.0 * L * C
        flops.append(F * n)
        byts.append(params[d] * es + act * n * es)
        cin, Lprev = C, L
    return dict(flops_by_depth=flops, bytes_by_depth=byts, params_by_depth=params, params_global=float(glob),
                flops=sum(flops), bytes=sum(byts) + glob * es, weight_bytes=(sum(params) + glob) * es)


def step_roofline_ms(work: Dict, peak_flops: float, peak_bytes_per_s: float) -> float:
    """sum_d max(F_d / P, Q_d / BW) in milliseconds."""
    t = sum(max(f / peak_flops, q / peak_bytes_per_s) for f, q in zip(work["flops_by_depth"], work["bytes_by_depth"]))
    t += work["params_global"] * (work["weight_bytes"] / max(sum(work["params_by_depth"]) + work["params_global"], 1.0)) / peak_bytes_per_s
    return t * 1e3


# R(2+1)D-18 onset net (main/onset_net.py, main/resnet.py): measured by forward hooks on the imported reference
ONSET_NET_GFLOP_PER_CLIP = 293.2
ONSET_NET_PARAMS = 31_365_918
