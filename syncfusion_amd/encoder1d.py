"""Host-side mirror of ``audio_encoders_pytorch.Encoder1d`` as the reference instantiates it.

``exp/model/diffusion.yaml:35-43``: ``Encoder1d(in_channels=1, channels=2, multipliers=[1,1,4,8,16,32,64,128,128],
factors=[1,4,4,4,2,2,2,2], num_blocks=[2]*8, resnet_groups=2, patch_size=1)``; the reference calls
``onsets_encoder(y, with_info=True)`` and slices ``info['xs'][2:-1]`` (main/generation.py:71,80,
main/module_diffusion.py:76,196).  The forward runs in the HIP engine (``sf_encoder1d_forward``);
the ``torch.nn`` parameters are the fp32 masters (SURVEY.md appendix A.4 for the structure).  With autograd recording
(the reference trains this encoder together with the U-Net, main/module_diffusion.py:53-61) the differentiable
composition of ``syncfusion_amd.training`` runs instead.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import _lib
from ._engine import EncoderEngine
from .diffusion import _conv_init, _register

Tensor = torch.Tensor


class Encoder1d(nn.Module):
    def __init__(self, in_channels: int, channels: int, multipliers: Sequence[int], factors: Sequence[int],
                 num_blocks: Sequence[int], patch_size: int = 1, resnet_groups: int = 8, out_channels: Optional[int] = None,
                 seed: Optional[int] = None):
        super().__init__()
        self.num_layers = len(multipliers) - 1
        assert len(factors) == self.num_layers and len(num_blocks) == self.num_layers
        assert patch_size == 1, "the reference config uses patch_size=1"
        assert out_channels is None, "the reference config leaves out_channels unset (to_out = Identity)"
        self.downsample_factor = patch_size
        for f in factors:
            self.downsample_factor *= f
        self.out_channels = out_channels
        self.hparams = dict(in_channels=in_channels, channels=channels, multipliers=list(multipliers), factors=list(factors),
                            num_blocks=list(num_blocks), resnet_groups=resnet_groups, patch_size=patch_size)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        c0 = channels * multipliers[0]
        self._res("to_in", in_channels, c0, gen)
        cin = c0
        for i, f in enumerate(factors):
            cout = channels * multipliers[i + 1]
            self._conv(f"downsamples.{i}.down", (cout, cin, 2 * f + 1), gen)
            for j in range(num_blocks[i]):
                self._res(f"downsamples.{i}.blocks.{j}", cout, cout, gen)
            cin = cout
        self._engine: Optional[EncoderEngine] = None

    def _conv(self, name, shape, gen):
        w, b = _conv_init(shape, gen)
        _register(self, name + ".weight", nn.Parameter(w))
        _register(self, name + ".bias", nn.Parameter(b))

    def _res(self, pre, cin, cout, gen):
        for blk, c_in in (("block1", cin), ("block2", cout)):
            _register(self, f"{pre}.{blk}.gn.weight", nn.Parameter(torch.ones(c_in)))
            _register(self, f"{pre}.{blk}.gn.bias", nn.Parameter(torch.zeros(c_in)))
            self._conv(f"{pre}.{blk}.conv", (cout, c_in, 3), gen)
        if cin != cout:
            self._conv(f"{pre}.to_out", (cout, cin, 1), gen)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def forward(self, x: Tensor, with_info: bool = False) -> Union[Tensor, Tuple[Tensor, Dict[str, List[Tensor]]]]:
        _lib.require_gpu_tensor(x, "Encoder1d.forward")
        from . import training

        if training.wants_grad(self, x):      # a training step: differentiable composition (syncfusion_amd/training.py)
            z, info = training.encoder1d_forward(self, x)
            return (z, info) if with_info else z
        with torch.no_grad():
            return self._forward_engine(x, with_info)

    def _forward_engine(self, x: Tensor, with_info: bool):
        if self._engine is None or self._engine.stale(self):
            self._engine = EncoderEngine(self)
        outs = self._engine.forward(x)
        xs = [x] + outs + [outs[-1]]          # [x, to_in, ds_0..ds_{n-1}, to_out(=Identity)]
        return (outs[-1], dict(xs=xs)) if with_info else outs[-1]
