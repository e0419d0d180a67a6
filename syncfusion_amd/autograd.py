"""Differentiable building blocks on the HIP kernels -- the first slice of the training step.

The reference trains ``DiffusionModel.forward`` (v-objective MSE, main/module_diffusion.py:73-82) in fp32
(exp/train_diffusion_gh.yaml:87).  The inference engine behind ``UNetV0.forward`` keeps no autograd graph; this module
provides, as ``torch.autograd.Function``s whose forward AND backward run in the HIP library (``sf_op_conv1d_cl`` /
``sf_op_conv1d_bwd_cl``), the two operations that carry ~90 % of the U-Net's parameters and FLOPs:

* ``gn_silu_conv1d(x, weight, bias, gamma, beta, groups, eps)``  --  ``Conv1d(SiLU(GroupNorm(x)))`` with stride 1 and
  "same" padding, the ResnetItem convolution (a-unet ResnetBlock);
* ``conv1d(x, weight, bias)``  --  plain stride-1 "same" Conv1d, e.g. the 1x1 InjectChannels convolution over ``cat[x, ctx]``.

Tensors are ``(B, C, L)`` fp32 CUDA tensors exactly as the reference's modules see them; the channels-last transposes around
the kernels are plumbing.  Everything else of a full training step (LayerNorm-modulate, attention, the modulation Linears,
the patchify / up convolutions) is not implemented yet: ``VDiffusion.forward`` still returns a loss without a graph.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

Tensor = torch.Tensor


def _cl(x: Tensor) -> Tensor:
    return x.transpose(1, 2).contiguous()


class _ConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor], groups: int, eps: float):
        _lib.require_gpu_tensor(x, "syncfusion_amd.autograd")
        lib = _lib.load()
        B, Cc, L = x.shape
        N, Cw, taps = weight.shape
        if Cw != Cc or taps % 2 != 1:
            raise ValueError(f"weight {tuple(weight.shape)} does not match input channels {Cc} (odd kernel sizes only)")
        pad = taps // 2
        c_real = Cc
        with torch.cuda.device(x.device):
            x_cl = _cl(_lib.f32c(x))
            w = _lib.f32c(weight)
            if groups == 0 and Cc % 32 != 0 and N > 32:
                # plain convolutions over odd channel counts (cat[x, ctx]) run on the MFMA kernels with the channels zero-padded
                # to a multiple of 32 (the engine pads the context buffer the same way); gradients are sliced back
                Cp = (Cc + 31) // 32 * 32
                x_cl = torch.nn.functional.pad(x_cl, (0, Cp - Cc))
                w = torch.nn.functional.pad(w, (0, 0, 0, Cp - Cc)).contiguous()
                Cc = Cp
            b = _lib.f32c(bias) if bias is not None else None
            g = _lib.f32c(gamma) if groups > 0 else None
            be = _lib.f32c(beta) if groups > 0 else None
            out = torch.empty(B, L, N, dtype=torch.float32, device=x.device)
            ws = torch.empty(max(256, 4 * N * Cc * taps + 8 * B * 64 * groups + (1 << 16)), dtype=torch.uint8, device=x.device)
            _lib.check(lib.sf_op_conv1d_cl(_lib.SF_F32, x_cl.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                           g.data_ptr() if g is not None else None, be.data_ptr() if be is not None else None, int(groups), float(eps),
                                           None, B, L, Cc, N, taps, 1, pad, 1, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)),
                       "sf_op_conv1d_cl")
        ctx.save_for_backward(x_cl, w, g if g is not None else x_cl.new_empty(0), be if be is not None else x_cl.new_empty(0))
        ctx.meta = (B, L, Cc, N, taps, pad, int(groups), float(eps), bias is not None, c_real)
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, dy: Tensor):
        lib = _lib.load()
        x_cl, w, g, be = ctx.saved_tensors
        B, L, Cc, N, taps, pad, groups, eps, has_bias, c_real = ctx.meta
        dev = x_cl.device
        with torch.cuda.device(dev):
            dy_cl = _cl(_lib.f32c(dy))
            dx = torch.empty_like(x_cl)
            dw = torch.empty_like(w)
            db = torch.empty(N, dtype=torch.float32, device=dev) if has_bias else None
            dgb = torch.empty(2 * Cc, dtype=torch.float32, device=dev) if groups > 0 else None
            n = lib.sf_op_conv1d_bwd_workspace_bytes(B, L, Cc, N, taps, groups)
            if n < 0:
                raise _lib.SyncFusionAmdError(lib.sf_last_error().decode())
            ws = torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.sf_op_conv1d_bwd_cl(x_cl.data_ptr(), w.data_ptr(), g.data_ptr() if groups > 0 else None, be.data_ptr() if groups > 0 else None,
                                               groups, eps, dy_cl.data_ptr(), B, L, Cc, N, taps, pad, dx.data_ptr(), dw.data_ptr(),
                                               db.data_ptr() if db is not None else None, dgb.data_ptr() if dgb is not None else None,
                                               ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)), "sf_op_conv1d_bwd_cl")
        if c_real != Cc:
            dx, dw = dx[:, :, :c_real], dw[:, :c_real]
        return (dx.transpose(1, 2), dw, db, dgb[:Cc] if dgb is not None else None, dgb[Cc:] if dgb is not None else None, None, None)


def gn_silu_conv1d(x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, groups: int, eps: float = 1e-5) -> Tensor:
    """``F.conv1d(F.silu(F.group_norm(x, groups, gamma, beta, eps)), weight, bias, padding=k//2)`` with HIP forward and backward."""
    return _ConvBlockFn.apply(x, weight, bias, gamma, beta, int(groups), float(eps))


def conv1d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """``F.conv1d(x, weight, bias, padding=k//2)`` (stride 1) with HIP forward and backward."""
    return _ConvBlockFn.apply(x, weight, bias, None, None, 0, 0.0)
