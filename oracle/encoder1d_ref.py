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


def _conv_block(P, pre, x, groups):
    h = F.group_norm(x, groups, P[pre + ".gn.weight"], P[pre + ".gn.bias"], eps=1e-5)
    return F.conv1d(F.silu(h), P[pre + ".conv.weight"], P[pre + ".conv.bias"], padding=1)


def _resnet_block(P, pre, x, groups):
    h = _conv_block(P, pre + ".block1", x, groups)
    h = _conv_block(P, pre + ".block2", h, groups)
    if (pre + ".to_out.weight") in P:
        x = F.conv1d(x, P[pre + ".to_out.weight"], P[pre + ".to_out.bias"])
    return h + x


def encoder1d_forward(P: Dict[str, Tensor], cfg, x: Tensor) -> Tuple[Tensor, Dict[str, List[Tensor]]]:
    """Encoder1d.forward(x, with_info=True) -> (z, {"xs": [x, to_in, ds_0..ds_{n-1}, to_out]})."""
    assert cfg["patch_size"] == 1
    xs = [x]
    x = _resnet_block(P, "to_in", x, 1)            # Patcher(patch_size=1) = ResnetBlock1d(groups=1)
    xs.append(x)
    for i, f in enumerate(cfg["factors"]):
        pre = f"downsamples.{i}"
        x = F.conv1d(x, P[pre + ".down.weight"], P[pre + ".down.bias"], stride=f, padding=f)
        for j in range(cfg["num_blocks"][i]):
            x = _resnet_block(P, f"{pre}.blocks.{j}", x, cfg["resnet_groups"])
        xs.append(x)
    xs.append(x)                                   # to_out = Identity (out_channels unset)
    return x, {"xs": xs}
