"""Host-side mirror of the reference's video onset encoder.

``VideoOnsetNet(pretrained).forward(x: (N,3,T,H,W)) -> (N,T)`` keeps the surface of
``main/onset_net.py:46-63`` (and therefore of ``cfg/model/model-onset.yaml:5-8``)
and exposes the *same* ``state_dict`` keys as the reference (226 tensors,
``net.model.stem.0.weight`` ... ``fc.2.bias``; main/resnet.py:195-286,
main/onset_net.py:12-38), so ``load_state_dict`` of a reference checkpoint works.

The ``torch.nn`` modules below only HOLD parameters.  ``forward`` never runs them:
it hands raw device pointers to the HIP engine behind the C ABI
(``sf_onsetnet_forward`` in include/syncfusion_amd.h), which folds the eval-mode
BatchNorms into the convolutions and runs every (1,k,k)/(3,1,1)/1x1x1 convolution
as an MFMA implicit GEMM on channels-last activations.  There is no CPU path: a
CPU tensor, or a missing extension, raises.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from ._engine import OnsetNetEngine

# (name, planes, spatial stride of the first block) -- main/resnet.py:215-222
_STAGES = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))


def midplanes(inplanes: int, planes: int) -> int:
    """Factorised (2+1)D bottleneck width, main/resnet.py:86-87."""
    return (inplanes * planes * 27) // (inplanes * 9 + 3 * planes)


def _conv(cin, cout, k, s=(1, 1, 1), p=(0, 0, 0)):
    return nn.Conv3d(cin, cout, kernel_size=k, stride=s, padding=p, bias=False)


def _r2plus1(cin, cout, mid, stride):
    # spatial (1,3,3) -> BN -> ReLU -> temporal (3,1,1) with stride 1 (main/onset_net.py:19-36)
    return nn.Sequential(_conv(cin, mid, (1, 3, 3), (1, stride, stride), (0, 1, 1)), nn.BatchNorm3d(mid),
                         nn.ReLU(inplace=True), _conv(mid, cout, (3, 1, 1), (1, 1, 1), (1, 0, 0)))


class _Residual(nn.Module):
    """Parameter holder with the reference BasicBlock's child names (conv1, conv2, downsample)."""

    def __init__(self, cin: int, planes: int, stride: int):
        super().__init__()
        self.conv1 = nn.Sequential(_r2plus1(cin, planes, midplanes(cin, planes), stride),
                                   nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(_r2plus1(planes, planes, midplanes(cin, planes), 1), nn.BatchNorm3d(planes))  # same midplanes as conv1 (main/resnet.py:86-98)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(_conv(cin, planes, (1, 1, 1), (1, stride, stride)), nn.BatchNorm3d(planes))
        self.stride = stride


class _Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(_conv(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), nn.BatchNorm3d(45), nn.ReLU(inplace=True),
                                  _conv(45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)), nn.BatchNorm3d(64), nn.ReLU(inplace=True))
        cin = 64
        for name, planes, stride in _STAGES:
            setattr(self, name, nn.Sequential(_Residual(cin, planes, stride), _Residual(planes, planes, 1)))
            cin = planes
        for m in self.modules():                      # main/resnet.py:273-286
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _KeepTemp(nn.Module):
    def __init__(self):
        super().__init__()
        self.model = _Trunk()


class VideoOnsetNet(nn.Module):
    """Drop-in for ``main.onset_net.VideoOnsetNet`` (HIP forward, inference only)."""

    def __init__(self, pretrained: bool = False, dtype: str = "fp32"):
        super().__init__()
        self.net = _KeepTemp()
        self.fc = nn.Sequential(nn.Linear(512, 128), nn.ReLU(True), nn.Linear(128, 1))
        self.compute_dtype = dtype
        self._engine: Optional[OnsetNetEngine] = None
        if pretrained:
            # the reference pulls Kinetics-400 weights from download.pytorch.org (main/resnet.py:291-294);
            # this build is offline-safe: point SYNCFUSION_R2PLUS1D_WEIGHTS at the same .pth file instead.
            path = os.environ.get("SYNCFUSION_R2PLUS1D_WEIGHTS")
            if not path:
                raise RuntimeError("pretrained=True needs SYNCFUSION_R2PLUS1D_WEIGHTS=<r2plus1d_18-91a641e6.pth> (no network access)")
            sd = torch.load(path, map_location="cpu")
            own = self.net.model.state_dict()
            self.net.model.load_state_dict({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}, strict=False)

    def _get_engine(self) -> OnsetNetEngine:
        if self._engine is None or self._engine.stale(self):
            self._engine = OnsetNetEngine(self, self.compute_dtype)
        return self._engine

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 5 or x.shape[1] != 3:
            raise ValueError(f"VideoOnsetNet expects (N, 3, T, H, W), got {tuple(x.shape)}")
        _lib.require_gpu_tensor(x, "VideoOnsetNet.forward")
        if self.training:
            raise RuntimeError("VideoOnsetNet (HIP) implements the eval-mode forward only (BatchNorm running stats); call .eval()")
        return self._get_engine().forward(x)
