"""Differentiable building blocks on the HIP kernels -- the operations of the training step.

The reference trains ``DiffusionModel.forward`` (v-objective MSE, main/module_diffusion.py:73-82) in fp32
(exp/train_diffusion_gh.yaml:87).  The inference engine behind ``UNetV0.forward`` keeps no autograd graph; this module
provides ``torch.autograd.Function``s whose forward AND backward run in the HIP library:

* ``gn_silu_conv1d(x, weight, bias, gamma, beta, groups, eps)``  --  ``Conv1d(SiLU(GroupNorm(x)))`` with stride 1 and
  "same" padding, the ResnetItem convolution (a-unet ResnetBlock)
  (``sf_op_conv1d_cl`` / ``sf_op_conv1d_bwd_cl``);
* ``conv1d(x, weight, bias)``  --  plain stride-1 "same" Conv1d: the 1x1 InjectChannels convolution over ``cat[x, ctx]``, the
  attention projections, and (after a reshape / gather) the patchify, strided and up-sampling convolutions;
* ``ln_modulate(x, scale_shift, eps)``  --  LayerNorm over the channels fused with the a-unet Modulation, also the affine
  pre-norm of the attention blocks (``sf_op_ln_modulate`` / ``sf_op_ln_modulate_bwd``);
* ``attention(q, kv, heads)``  --  multi-head softmax attention, head dim 64 (``sf_op_attention`` / ``sf_op_attention_bwd``).

fp32 CUDA tensors; the convolutions take ``(B, C, L)`` as the reference's modules see them or, with ``channels_last=True``,
the kernels' own ``(B, L, C)`` rows.  ``syncfusion_amd/training.py`` composes the U-Net and the onset encoder from these.
No atomics anywhere: a second backward gives the same bits.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

from . import _lib

Tensor = torch.Tensor


def _cl(x: Tensor) -> Tensor:
    return x.transpose(1, 2).contiguous()


# Arithmetic of the training step's GEMMs (forward, data gradient, weight gradient): "fp32x" = fp32 tensors, every product built from split
# fp16 operands (three 16-bit MFMAs, fp32 accumulation: 7.5e-8 rel-L2 against fp64 on a K = 3072 GEMM, inside plain fp32 MFMA's 3.5e-7) at
# 2.5x the fp32 matrix rate; "fp32" = v_mfma_f32_32x32x2_f32.  The reference trains with `precision: 32` (exp/train_diffusion_gh.yaml:87);
# both settings meet the gradient tests' unchanged tolerances.
GEMM_DTYPE = os.environ.get("SF_TRAIN_GEMM", "fp32x")

_SELF_PACK = os.environ.get("SF_TRAIN_SELF_PACK") == "1"  # A/B aid: the backward pass packs the data-gradient weight images itself (one more launch per convolution)
_FUSED_GN = os.environ.get("SF_TRAIN_FUSED_GN") == "1"   # A/B aid: GroupNorm+SiLU as the convolution kernel's prologue, recomputed in backward


def _capturing(t: Tensor) -> bool:
    """is the current stream of ``t``'s device recording a HIP graph?  (The plan is neither recorded nor built then: building uploads a table.)"""
    return bool(t.is_cuda and torch.cuda.is_current_stream_capturing())


class PackPlan:
    """The weight images of every convolution of one training forward, written by ONE launch at its start (``sf_train_pack_many``) instead of
    one launch per convolution (240 launches of ~8 us per step, mostly launch latency).

    ``with plan:`` around a forward pass.  The FIRST pass records which weights are convolved at which geometry (and packs per convolution,
    as without a plan); at its end the plan allocates one persistent buffer for all images and a descriptor table in device memory.  Every
    LATER pass packs all of them from the CURRENT weight values with one launch on entry -- nothing is cached across passes, so an optimizer
    step, ``load_state_dict`` or any in-place update between passes is always seen -- and the convolutions read their images from the
    buffer.  A convolution the plan does not know (other shapes, a temporary weight) packs its own as before; any miss makes the plan
    re-record on the next pass.  The plan keeps the recorded weight tensors alive, so the addresses in its table stay valid.  Cost: the
    buffer holds two 4-byte images per weight element (1.6 GB for the 215 M-parameter U-Net) per plan; ``training._pack_plan`` keeps at
    most four plans (input shapes) per module.  ``SF_TRAIN_SELF_PACK=1`` turns plans (and the fused per-convolution pack) off."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.state, self.items, self.buf, self.desc, self.total_tiles, self.misses, self.n_packed, self.hits = "record", {}, None, None, 0, 0, 0, 0

    @staticmethod
    def _key(w, N, C, taps):
        return (w.data_ptr(), int(N), int(C), int(taps))

    def record(self, w, geom, mask, need_dg):
        if self.state == "record" and not _capturing(w):
            self.items.setdefault(self._key(w, geom[3], geom[2], geom[4]), dict(w=w, geom=tuple(geom), mask=int(mask), need_dg=bool(need_dg)))

    def lookup(self, w, geom, need_dg):
        if self.state != "ready":
            return None
        it = self.items.get(self._key(w, geom[3], geom[2], geom[4]))
        if it is None or it["geom"] != tuple(geom) or (need_dg and not it["need_dg"]):
            self.misses += 1
            return None
        if it["mask"] < 0:   # recorded as a convolution that packs its own images -- not a miss
            return None
        self.hits += 1
        return it

    def __enter__(self):
        self._outer, _TLS.plan = getattr(_TLS, "plan", None), self
        self.hits = 0
        if self.state == "ready":
            _lib.check(_lib.load().sf_train_pack_many(self.desc.data_ptr(), self.n_packed, self.total_tiles, _lib.stream_ptr(self.buf.device)),
                       "sf_train_pack_many")
        return self

    def __exit__(self, *exc):
        _TLS.plan = self._outer
        if exc[0] is not None:
            self.reset()
        elif self.state == "ready" and (self.misses or not self.hits):   # an unknown convolution, or a pass that used none of the images
            self.reset()                                                   # (weights re-created, cast on the fly): record again
        elif self.state == "record" and self.items and not _capturing(next(iter(self.items.values()))["w"]):
            self._finalize()
        return False

    def _finalize(self):
        packed = [it for it in self.items.values() if it["mask"] >= 0]
        if not packed:
            return
        dev = packed[0]["w"].device
        off, tiles, rows = 0, 0, []

        def take(nbytes):
            nonlocal off
            o, off = off, off + (nbytes + 255) // 256 * 256
            return o

        for it in packed:
            B, L, C, N, taps, pad, gr = it["geom"]
            m, nk = it["mask"], 4 * N * C * taps
            it["fw_off"] = take(nk) if m & 1 else None
            it["fwx_off"] = take(nk) if m & 2 else None
            it["dg_off"] = take(2 * nk) if it["need_dg"] else None   # [dg fp32 | dgx], as sf_op_conv1d_bwd_cl_p reads them
            it["tile0"] = tiles
            tiles += ((C + 31) // 32) * ((N + 31) // 32)
        self.buf = torch.empty(max(off, 256), dtype=torch.uint8, device=dev)
        base = self.buf.data_ptr()
        for it in packed:
            B, L, C, N, taps, pad, gr = it["geom"]
            m, nk = it["mask"], 4 * N * C * taps
            dg = base + it["dg_off"] if it["need_dg"] and m & 4 else 0
            dgx = base + it["dg_off"] + nk if it["need_dg"] and m & 8 else 0
            rows.append([it["w"].data_ptr(), base + it["fw_off"] if m & 1 else 0, base + it["fwx_off"] if m & 2 else 0, dg, dgx, N | (C << 32),
                         taps | (it["tile0"] << 32)])
        self.desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.total_tiles, self.n_packed, self.state, self.misses = tiles, len(packed), "ready", 0


_TLS = threading.local()   # .plan: the PackPlan of the forward pass running on THIS thread (None outside ``with plan:``)


class _ConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor], groups: int, eps: float,
                channels_last: bool = False, residual: Optional[Tensor] = None, passthrough: bool = False):
        _lib.require_gpu_tensor(x, "syncfusion_amd.autograd")
        ctx.set_materialize_grads(False)
        ctx.passthrough = bool(passthrough)
        lib = _lib.load()
        if channels_last:
            B, L, Cc = x.shape
        else:
            B, Cc, L = x.shape
        N, Cw, taps = weight.shape
        if Cw != Cc or taps % 2 != 1:
            raise ValueError(f"weight {tuple(weight.shape)} does not match input channels {Cc} (odd kernel sizes only)")
        pad = taps // 2
        c_real, n_real = Cc, N
        if groups > 0 and Cc % 32 != 0 and (Cc > 32 or N > 32):
            raise ValueError(f"gn_silu_conv1d: {Cc} -> {N} channels: channel counts above 32 must be multiples of 32")
        with torch.cuda.device(x.device):
            x_cl = _lib.f32c(x) if channels_last else _cl(_lib.f32c(x))
            w = _lib.f32c(weight)
            if groups == 0 and Cc % 32 != 0 and N > 32:
                # plain convolutions over odd channel counts (cat[x, ctx]) run on the MFMA kernels with the channels zero-padded
                # to a multiple of 32 (the engine pads the context buffer the same way); gradients are sliced back
                Cp = (Cc + 31) // 32 * 32
                x_cl = torch.nn.functional.pad(x_cl, (0, Cp - Cc))
                w = torch.nn.functional.pad(w, (0, 0, 0, Cp - Cc)).contiguous()
                Cc = Cp
            if groups == 0 and N % 32 != 0 and Cc > 32:
                # ... and thin outputs of wide inputs get zero output channels up to a multiple of 32 (their dgrad runs on MFMA)
                Np = (N + 31) // 32 * 32
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, Np - N)).contiguous()
                bias = torch.nn.functional.pad(bias, (0, Np - N)) if bias is not None else None
                N = Np
            b = _lib.f32c(bias) if bias is not None else None
            g = _lib.f32c(gamma) if groups > 0 else None
            be = _lib.f32c(beta) if groups > 0 else None
            out = torch.empty(B, L, N, dtype=torch.float32, device=x.device)
            ws = torch.empty(max(256, 12 * N * Cc * taps + 8 * B * 64 * groups + (1 << 16)), dtype=torch.uint8, device=x.device)
            # residual: added in the convolution's epilogue (one pass less over the tensor than a separate add); its gradient is dy itself
            res_cl, res_late = None, None
            if residual is not None:
                res_cl = _lib.f32c(residual) if channels_last else _cl(_lib.f32c(residual))
                if tuple(res_cl.shape) != (B, L, n_real):
                    raise ValueError(f"residual shape {tuple(residual.shape)} does not match the output")
                if n_real != N:   # zero-padded output width: the kernel's rows are wider than the residual's; add afterwards
                    res_cl, res_late = None, res_cl
            # MFMA-path GroupNorm convolutions: a = silu(groupnorm(x)) is materialised once and KEPT for the backward pass, and the convolution
            # itself runs as a plain GEMM (macro tiles on the wide levels) -- the fused-prologue kernel was the slowest forward GEMM of the
            # step and the backward pass recomputed a for the weight gradient (SF_TRAIN_FUSED_GN=1: the former path).
            act, stats = None, None
            if groups > 0 and Cc % 32 == 0 and not _FUSED_GN:
                act = torch.empty_like(x_cl)
                nst = int(lib.sf_op_gn_silu_train_stats_floats(B, L, Cc, int(groups)))
                if nst < 0:
                    raise _lib.SyncFusionAmdError(lib.sf_last_error().decode())
                stats = torch.empty(nst, dtype=torch.float32, device=x.device)   # the chunk statistics the GroupNorm backward reads (may be empty)
                _lib.check(lib.sf_op_gn_silu_train(x_cl.data_ptr(), g.data_ptr(), be.data_ptr(), int(groups), float(eps), B, L, Cc, act.data_ptr(),
                                                   stats.data_ptr() if nst > 0 else None, _lib.stream_ptr(x.device)), "sf_op_gn_silu_train")
            src, gr = (act, 0) if act is not None else (x_cl, int(groups))
            # the weight images of the backward pass's data-gradient GEMM come out of the SAME pack launch as the forward images (one launch
            # per weight and step; the backward pass packs nothing), when a data gradient will be asked for
            dgp = None
            need_dg = taps <= 9 and (groups > 0 or ctx.needs_input_grad[0]) and not _SELF_PACK
            plan, geom = getattr(_TLS, "plan", None), (B, L, Cc, N, taps, pad, gr)
            # (only weights that live across passes: a parameter or a view of one -- a matrix computed in this pass has a new address every time)
            stable_w = weight.is_leaf or (weight._base is not None and weight._base.is_leaf)
            plannable = plan is not None and not _SELF_PACK and c_real == Cc and n_real == N and taps <= 9 and stable_w and w.data_ptr() == weight.data_ptr()
            it = plan.lookup(w, geom, need_dg) if plannable else None
            if it is not None:   # images written by the plan's one launch at the start of this pass
                base = plan.buf.data_ptr()
                if need_dg:
                    dgp = plan.buf[it["dg_off"]:it["dg_off"] + 8 * N * Cc * taps]
                _lib.check(lib.sf_op_conv1d_train_fwd_pk(_lib.DTYPES[GEMM_DTYPE], src.data_ptr(), w.data_ptr(),
                                                         base + it["fw_off"] if it["fw_off"] is not None else None,
                                                         base + it["fwx_off"] if it["fwx_off"] is not None else None,
                                                         b.data_ptr() if b is not None else None, g.data_ptr() if gr > 0 else None,
                                                         be.data_ptr() if gr > 0 else None, gr, float(eps), res_cl.data_ptr() if res_cl is not None else None,
                                                         B, L, Cc, N, taps, pad, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)),
                           "sf_op_conv1d_train_fwd_pk")
            else:
                if plannable and plan.state == "record":
                    mask = int(lib.sf_op_conv1d_train_images(_lib.DTYPES[GEMM_DTYPE], w.data_ptr(), B, L, Cc, N, taps, pad, gr))
                    plan.record(w, geom, -1 if mask < 0 or mask & 16 else mask, need_dg)   # (-1: this launch keeps packing its own)
                if need_dg:
                    dgp = torch.empty(int(lib.sf_op_conv1d_dgrad_pack_bytes(Cc, N, taps)), dtype=torch.uint8, device=x.device)
                _lib.check(lib.sf_op_conv1d_train_fwd(_lib.DTYPES[GEMM_DTYPE], src.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                                      g.data_ptr() if gr > 0 else None, be.data_ptr() if gr > 0 else None, gr, float(eps),
                                                      res_cl.data_ptr() if res_cl is not None else None, B, L, Cc, N, taps, pad, out.data_ptr(),
                                                      dgp.data_ptr() if dgp is not None else None, dgp.numel() if dgp is not None else 0, ws.data_ptr(),
                                                      ws.numel(), _lib.stream_ptr(x.device)),
                           "sf_op_conv1d_train_fwd")
        ctx.save_for_backward(x_cl, w, g if g is not None else x_cl.new_empty(0), be if be is not None else x_cl.new_empty(0),
                              act if act is not None else x_cl.new_empty(0), stats if stats is not None else x_cl.new_empty(0),
                              dgp if dgp is not None else x_cl.new_empty(0))
        ctx.meta = (B, L, Cc, N, taps, pad, int(groups), float(eps), bias is not None, c_real, n_real, bool(channels_last))
        if n_real != N:
            out = out[:, :, :n_real]
            if res_late is not None:
                out = out + res_late
        out = out if channels_last else out.transpose(1, 2)
        # passthrough: x again as a second output.  A residual connection that bypasses this op reads THAT tensor, so both gradients of x
        # arrive in this node's backward and are summed inside the GroupNorm backward's pass (no element-wise launch by the autograd engine)
        return (out, x.view_as(x)) if passthrough else out

    @staticmethod
    def backward(ctx, dy: Optional[Tensor], d_pass: Optional[Tensor] = None):
        lib = _lib.load()
        x_cl, w, g, be, act, stats, dgp = ctx.saved_tensors
        B, L, Cc, N, taps, pad, groups, eps, has_bias, c_real, n_real, channels_last = ctx.meta
        dev = x_cl.device
        if dy is None:   # only the passthrough output was used
            return (d_pass,) + (None,) * 9
        add = None
        if d_pass is not None:
            add = _lib.f32c(d_pass) if channels_last else _cl(_lib.f32c(d_pass))
        fused_add = add is not None and groups > 0 and c_real == Cc
        with torch.cuda.device(dev):
            dy_cl = _lib.f32c(dy) if channels_last else _cl(_lib.f32c(dy))
            if n_real != N:
                dy_cl = torch.nn.functional.pad(dy_cl, (0, N - n_real))
            # ctx.needs_input_grad: (x, weight, bias, gamma, beta, ...).  The data gradient of a convolution without a GroupNorm in
            # front and the weight gradient of a frozen layer are not computed at all (the C ABI takes NULL for them).
            need_dx = ctx.needs_input_grad[0] or groups > 0
            need_dw = ctx.needs_input_grad[1]
            dx = torch.empty_like(x_cl) if need_dx else None
            dw = torch.empty_like(w) if need_dw else None
            db = torch.empty(N, dtype=torch.float32, device=dev) if has_bias else None
            dgb = torch.empty(2 * Cc, dtype=torch.float32, device=dev) if groups > 0 else None
            n = lib.sf_op_conv1d_bwd_workspace_bytes(B, L, Cc, N, taps, groups)
            if n < 0:
                raise _lib.SyncFusionAmdError(lib.sf_last_error().decode())
            ws = torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)
            tail = (B, L, Cc, N, taps, pad, dx.data_ptr() if dx is not None else None,
                    dw.data_ptr() if dw is not None else None, db.data_ptr() if db is not None else None, dgb.data_ptr() if dgb is not None else None,
                    ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
            gp, bp = (g.data_ptr() if groups > 0 else None), (be.data_ptr() if groups > 0 else None)
            have_act = groups > 0 and act.numel() > 0
            _lib.check(lib.sf_op_conv1d_bwd_cl_p(_lib.DTYPES[GEMM_DTYPE], x_cl.data_ptr(), act.data_ptr() if have_act else None,
                                                 stats.data_ptr() if have_act and stats.numel() > 0 else None, w.data_ptr(),
                                                 dgp.data_ptr() if dgp.numel() > 0 else None, gp, bp, groups, eps, dy_cl.data_ptr(),
                                                 add.data_ptr() if fused_add else None, *tail),
                       "sf_op_conv1d_bwd_cl_p")
        if c_real != Cc:
            dx = dx[:, :, :c_real] if dx is not None else None
            dw = dw[:, :c_real] if dw is not None else None
        if n_real != N:
            dw, db = (dw[:n_real] if dw is not None else None), (db[:n_real] if db is not None else None)
        if add is not None and not fused_add and dx is not None:
            dx = dx + add
        if dx is not None and not channels_last:
            dx = dx.transpose(1, 2)
        if not ctx.needs_input_grad[0]:
            dx = None
        return (dx, dw, db, dgb[:Cc] if dgb is not None else None, dgb[Cc:] if dgb is not None else None,
                None, None, None, dy if ctx.needs_input_grad[8] else None, None)


def gn_silu_conv1d(x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Tensor, beta: Tensor, groups: int, eps: float = 1e-5,
                   channels_last: bool = False, residual: Optional[Tensor] = None, passthrough: bool = False):
    """``F.conv1d(F.silu(F.group_norm(x, groups, gamma, beta, eps)), weight, bias, padding=k//2) (+ residual)`` with HIP forward and
    backward.  ``channels_last``: x, residual and the result are ``(B, L, C)`` instead of ``(B, C, L)`` (no transposes around the kernels).
    ``passthrough``: returns ``(y, x')`` with ``x'`` = x; a residual branch around the op that reads ``x'`` has its gradient added to this
    op's dx inside the GroupNorm backward kernel (one pass over the tensor less than the autograd engine's separate add)."""
    return _ConvBlockFn.apply(x, weight, bias, gamma, beta, int(groups), float(eps), bool(channels_last), residual, bool(passthrough))


def conv1d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, channels_last: bool = False, residual: Optional[Tensor] = None) -> Tensor:
    """``F.conv1d(x, weight, bias, padding=k//2) (+ residual)`` (stride 1) with HIP forward and backward."""
    return _ConvBlockFn.apply(x, weight, bias, None, None, 0, 0.0, bool(channels_last), residual, False)


def length_sums(x: Tensor, y: Optional[Tensor] = None) -> Tensor:
    """``(x * y).sum(dim=1)`` (``y`` optional) of channels-last fp32 ``(B, L, C)`` tensors -> ``(B, C)``: the length reductions in the
    backward of a per-clip broadcast add and of the SkipModulate scale, in one HIP pass (no gradient is recorded: backward use only)."""
    _lib.require_gpu_tensor(x, "syncfusion_amd.autograd")
    lib = _lib.load()
    B, L, Cc = x.shape
    with torch.cuda.device(x.device):
        xc = _lib.f32c(x)
        yc = _lib.f32c(y) if y is not None else None
        if yc is not None and tuple(yc.shape) != (B, L, Cc):
            raise ValueError(f"length_sums: shapes {tuple(x.shape)} and {tuple(y.shape)} differ")
        n = lib.sf_op_length_sums_workspace_bytes(B, L, Cc)
        if n < 0:
            raise _lib.SyncFusionAmdError(lib.sf_last_error().decode())
        ws = torch.empty(max(int(n), 256), dtype=torch.uint8, device=x.device)
        out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        _lib.check(lib.sf_op_length_sums(xc.data_ptr(), yc.data_ptr() if yc is not None else None, B, L, Cc, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr(x.device)), "sf_op_length_sums")
    return out


class _LnModulateFn(torch.autograd.Function):
    """y = LayerNorm_C(x; eps, no affine) * (1 + ss[:, :C]) + ss[:, C:] on channels-last ``(B, L, C)`` rows (ss: ``(B, 2C)`` or None)."""

    @staticmethod
    def forward(ctx, x: Tensor, ss: Optional[Tensor], eps: float, passthrough: bool = False):
        _lib.require_gpu_tensor(x, "syncfusion_amd.autograd")
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        B, L, Cc = x.shape
        with torch.cuda.device(x.device):
            xc = _lib.f32c(x)
            sc = _lib.f32c(ss) if ss is not None else None
            if sc is not None and sc.data_ptr() % 16:   # a view into a wider buffer (one Linear for every Modulation): the kernel reads 16-byte vectors
                sc = sc.clone()
            out = torch.empty_like(xc)
            _lib.check(lib.sf_op_ln_modulate(_lib.SF_F32, xc.data_ptr(), sc.data_ptr() if sc is not None else None, float(eps), B, L, Cc, out.data_ptr(),
                                             _lib.stream_ptr(x.device)), "sf_op_ln_modulate")
        ctx.save_for_backward(xc, sc if sc is not None else xc.new_empty(0))
        ctx.meta = (B, L, Cc, float(eps), ss is not None)
        return (out, x.view_as(x)) if passthrough else out   # (passthrough: see gn_silu_conv1d)

    @staticmethod
    def backward(ctx, dy: Optional[Tensor], d_pass: Optional[Tensor] = None):
        lib = _lib.load()
        xc, sc = ctx.saved_tensors
        B, L, Cc, eps, has_ss = ctx.meta
        dev = xc.device
        if dy is None:   # only the passthrough output was used
            return d_pass, None, None, None
        with torch.cuda.device(dev):
            dyc = _lib.f32c(dy)
            add = _lib.f32c(d_pass) if d_pass is not None else None
            dx = torch.empty_like(xc)
            dss = torch.empty(B, 2 * Cc, dtype=torch.float32, device=dev) if has_ss else None
            ws = torch.empty(max(int(lib.sf_op_ln_modulate_bwd_workspace_bytes(B, L, Cc)), 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.sf_op_ln_modulate_bwd_add(xc.data_ptr(), sc.data_ptr() if has_ss else None, dyc.data_ptr(),
                                                     add.data_ptr() if add is not None else None, eps, B, L, Cc, dx.data_ptr(),
                                                     dss.data_ptr() if dss is not None else None, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)),
                       "sf_op_ln_modulate_bwd_add")
        return dx, dss, None, None


def ln_modulate(x: Tensor, scale_shift: Optional[Tensor], eps: float, passthrough: bool = False):
    """Modulation / pre-norm LayerNorm on ``(B, L, C)`` rows with HIP forward and backward.  An affine LayerNorm ``LN(x) * g + b``
    is ``ln_modulate(x, cat[g - 1, b] broadcast over the clips, eps)``.  ``passthrough``: returns ``(y, x')`` (see ``gn_silu_conv1d``)."""
    return _LnModulateFn.apply(x, scale_shift, float(eps), bool(passthrough))


class _AttentionFn(torch.autograd.Function):
    """Multi-head softmax attention (head dim 64) on packed projections: q ``(B, L, H*64)``, kv ``(B, L, 2*H*64)`` -> ``(B, L, H*64)``.
    The forward pass keeps the log-sum-exp of the scaled scores (B, H, L) for the backward pass (one score pass less there)."""

    @staticmethod
    def forward(ctx, q: Tensor, kv: Tensor, heads: int):
        _lib.require_gpu_tensor(q, "syncfusion_amd.autograd")
        lib = _lib.load()
        B, L, HD = q.shape
        H = int(heads)
        with torch.cuda.device(q.device):
            qc, kc = _lib.f32c(q), _lib.f32c(kv)
            out = torch.empty_like(qc)
            lse = torch.empty(B, H, L, dtype=torch.float32, device=q.device)
            _lib.check(lib.sf_op_attention_fwd_lse_x(_lib.DTYPES[GEMM_DTYPE], qc.data_ptr(), kc.data_ptr(), B, L, H, HD // H, out.data_ptr(), lse.data_ptr(),
                                                     _lib.stream_ptr(q.device)), "sf_op_attention_fwd_lse_x")
        ctx.save_for_backward(qc, kc, out, lse)
        ctx.meta = (B, L, H, HD // H)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        lib = _lib.load()
        qc, kc, out, lse = ctx.saved_tensors
        B, L, H, D = ctx.meta
        dev = qc.device
        with torch.cuda.device(dev):
            doc = _lib.f32c(dout)
            dq, dkv = torch.empty_like(qc), torch.empty_like(kc)
            ws = torch.empty(B * H * L * 4 + 256, dtype=torch.uint8, device=dev)
            _lib.check(lib.sf_op_attention_bwd_lse_x(_lib.DTYPES[GEMM_DTYPE], qc.data_ptr(), kc.data_ptr(), out.data_ptr(), doc.data_ptr(), lse.data_ptr(), B, L, H,
                                                     D, dq.data_ptr(), dkv.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)),
                       "sf_op_attention_bwd_lse_x")
        return dq, dkv, None


def attention(q: Tensor, kv: Tensor, heads: int) -> Tensor:
    """softmax(q k^T / sqrt(64)) v per head, HIP forward and backward."""
    return _AttentionFn.apply(q, kv, int(heads))
