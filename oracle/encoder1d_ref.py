"""Oracle: ``audio_encoders_pytorch.Encoder1d`` (==0.0.22) restated on the CPU.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED (package
absent; requirements.txt:24).  Follows SURVEY.md appendix A.4 and the
reference's config ``exp/model/diffusion.yaml:35-43``; callers
main/generation.py:71 and main/module_diffusion.py:76,196 slice ``xs[2:-1]``.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

DEFAULT_CONFIG = dict(
    in_channels=1,
    channels=2,
    multipliers=[1, 1, 4, 8, 16, 32, 64, 128, 128],
    factors=[1, 4, 4, 4, 2, 2, 2, 2],
    num_blocks=[2, 2, 2, 2, 2, 2, 2, 2],
    resnet_groups=2,
    patch_size=1,
)


# [RECALLED] facts about audio_encoders_pytorch that shapes cannot settle; switches as in oracle/unet_ref.py (cfg["variants"]),
# decided numerically by tools/pin_upstream.py.
RECALLED_DEFAULTS = dict(
    # (Patcher(patch_size=1) = ResnetBlock1d(num_groups=1): the only group count one input channel admits; a different `to_in`
    #  form -- a bare convolution, another kernel -- changes tensor SHAPES and fails the key map loudly)
    block_act="silu",        # ConvBlock1d activation: SiLU (default) | "relu"
)


def recalled_variants(cfg) -> Dict:
    v = dict(RECALLED_DEFAULTS)
    v.update(cfg.get("variants") or {})
    unknown = set(v) - set(RECALLED_DEFAULTS)
    assert not unknown, f"unknown oracle variant switches: {sorted(unknown)}"
    return v


def _conv_block(P, pre, x, groups, act="silu"):
    h = F.group_norm(x, groups, P[pre + ".gn.weight"], P[pre + ".gn.bias"], eps=1e-5)
    return F.conv1d(F.silu(h) if act == "silu" else F.relu(h), P[pre + ".conv.weight"], P[pre + ".conv.bias"], padding=1)


def _resnet_block(P, pre, x, groups, act="silu"):
    h = _conv_block(P, pre + ".block1", x, groups, act)
    h = _conv_block(P, pre + ".block2", h, groups, act)
    if (pre + ".to_out.weight") in P:
        x = F.conv1d(x, P[pre + ".to_out.weight"], P[pre + ".to_out.bias"])
    return h + x


def encoder1d_forward(P: Dict[str, Tensor], cfg, x: Tensor) -> Tuple[Tensor, Dict[str, List[Tensor]]]:
    """Encoder1d.forward(x, with_info=True) -> (z, {"xs": [x, to_in, ds_0..ds_{n-1}, to_out]})."""
    assert cfg["patch_size"] == 1
    var = recalled_variants(cfg)
    xs = [x]
    x = _resnet_block(P, "to_in", x, 1, var["block_act"])    # Patcher(patch_size=1) = ResnetBlock1d(groups=1)
    xs.append(x)
    for i, f in enumerate(cfg["factors"]):
        pre = f"downsamples.{i}"
        # (kernel 2f+1 / padding f is settled by the checkpoint's weight SHAPE: a different kernel fails the key map loudly)
        x = F.conv1d(x, P[pre + ".down.weight"], P[pre + ".down.bias"], stride=f, padding=f)
        for j in range(cfg["num_blocks"][i]):
            x = _resnet_block(P, f"{pre}.blocks.{j}", x, cfg["resnet_groups"], var["block_act"])
        xs.append(x)
    xs.append(x)                                   # to_out = Identity (out_channels unset)
    return x, {"xs": xs}
