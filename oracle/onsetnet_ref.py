"""Oracle: the reference's ``VideoOnsetNet`` (R(2+1)D-18, temporal stride removed) on the CPU.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PINNED: this restatement is
checked against the reference itself (main/onset_net.py:12-63 +
main/resnet.py:36-56,81-114,177-192,234-251, imported in the build container by
``oracle/gen_golden_onsetnet.py``) and against the committed vectors in
``tests/golden/onsetnet_*.npz``.

It consumes the reference's own ``state_dict`` layout (226 tensors,
``net.model.stem.0.weight`` ... ``fc.2.bias``), eval-mode BatchNorm (running
statistics), written with ``torch.nn.functional`` only.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

LAYERS = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))


def _bn(P, pre, x):
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"],
                        P[pre + ".weight"], P[pre + ".bias"], training=False, eps=1e-5)


def _conv2plus1d(P, pre, x, stride):
    # main/resnet.py:43-52 with the temporal stride forced to 1 (main/onset_net.py:19-36)
    x = F.conv3d(x, P[pre + ".0.weight"], None, stride=(1, stride, stride), padding=(0, 1, 1))
    x = F.relu(_bn(P, pre + ".1", x))
    return F.conv3d(x, P[pre + ".3.weight"], None, stride=(1, 1, 1), padding=(1, 0, 0))


def _basic_block(P, pre, x, stride, taps=None):
    # main/resnet.py:100-114
    res = x
    out = F.relu(_bn(P, pre + ".conv1.1", _conv2plus1d(P, pre + ".conv1.0", x, stride)))
    out = _bn(P, pre + ".conv2.1", _conv2plus1d(P, pre + ".conv2.0", out, 1))
    if (pre + ".downsample.0.weight") in P:
        res = _bn(P, pre + ".downsample.1",
                  F.conv3d(x, P[pre + ".downsample.0.weight"], None, stride=(1, stride, stride)))
    return F.relu(out + res)


def onsetnet_forward(P: Dict[str, Tensor], x: Tensor, taps: Optional[dict] = None) -> Tensor:
    """VideoOnsetNet.forward: (N, 3, T, H, W) -> (N, T) raw logits (main/onset_net.py:57-63)."""
    m = "net.model."
    # R2Plus1dStem (main/resnet.py:181-192)
    x = F.conv3d(x, P[m + "stem.0.weight"], None, stride=(1, 2, 2), padding=(0, 3, 3))
    x = F.relu(_bn(P, m + "stem.1", x))
    x = F.conv3d(x, P[m + "stem.3.weight"], None, stride=(1, 1, 1), padding=(1, 0, 0))
    x = F.relu(_bn(P, m + "stem.4", x))
    if taps is not None:
        taps["stem"] = x
    for name, _planes, stride in LAYERS:
        x = _basic_block(P, f"{m}{name}.0", x, stride)
        x = _basic_block(P, f"{m}{name}.1", x, 1)
        if taps is not None:
            taps[name] = x
    x = x.mean(dim=(3, 4))                  # AdaptiveAvgPool3d((None, 1, 1)) + squeeze -> (N, 512, T)
    x = x.transpose(-1, -2)                 # (N, T, 512)
    x = F.relu(F.linear(x, P["fc.0.weight"], P["fc.0.bias"]))
    x = F.linear(x, P["fc.2.weight"], P["fc.2.bias"])
    return x.squeeze(-1)


def onsetnet_flops(T: int, H: int, W: int) -> float:
    """MAC*2 of every conv + fc for one clip (matches SURVEY section 8a-8: 293.2 GFLOP at 30x112x112)."""
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    fl = 2.0 * T * Ho * Wo * 45 * 3 * 49 + 2.0 * T * Ho * Wo * 64 * 45 * 3
    cin = 64
    for _name, planes, stride in LAYERS:
        for blk in range(2):
            s = stride if blk == 0 else 1
            inp = cin if blk == 0 else planes
            Hn, Wn = (Ho + s - 1) // s, (Wo + s - 1) // s
            mid1 = (inp * planes * 27) // (inp * 9 + 3 * planes)
            mid2 = mid1  # one midplanes per BasicBlock (main/resnet.py:86-98)
            fl += 2.0 * T * Hn * Wn * (mid1 * inp * 9 + planes * mid1 * 3)
            fl += 2.0 * T * Hn * Wn * (mid2 * planes * 9 + planes * mid2 * 3)
            if blk == 0 and (s != 1 or inp != planes):
                fl += 2.0 * T * Hn * Wn * planes * inp
            Ho, Wo = Hn, Wn
        cin = planes
    fl += 2.0 * T * (512 * 128 + 128)
    return fl
