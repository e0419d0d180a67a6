"""Thin Python owners of the C-ABI engine handles (weights snapshot, workspace, stream plumbing)."""
from __future__ import annotations

import ctypes as C
import operator as _operator
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SF_MAX_DEPTH, EncoderConfig, UnetConfig, check


# Staleness of the packed weights, scoped to the module an engine was built from (nothing process-wide: no torch hook, no patched
# `Module._apply`).  An engine remembers every SLOT of its module tree -- (owner dict, name, object) for each parameter, buffer and
# submodule, and the size of each owner dict -- and a call checks that the slots still hold the same objects (`.to()` / `_apply` replace
# buffers out of place, `del m.w`, `m.w = Parameter(...)`, `load_state_dict(assign=True)`, `add_module`, direct `_buffers` writes all show up
# as a changed slot or a changed dict size) and that (data_ptr, _version) of the tensors are unchanged (in-place updates: optimizer steps,
# `load_state_dict`).  For the 215 M-parameter U-Net (842 tensors, ~300 modules) that is ~0.25 ms of host time per sample() call -- the
# full `module.parameters()` walk it replaces costs 1.4 ms -- and every kind of edit is seen on the NEXT call, not up to 16 calls later.
_REWALK_EVERY = 16     # calls between unconditional re-walks (a backstop: no known edit needs it)


class _ParamsVersion:
    """Slots and (data_ptr, _version) of every parameter and buffer of one module tree."""

    def __init__(self, module: torch.nn.Module):
        self._walk(module)
        self.value = self._read()

    def _walk(self, module: torch.nn.Module) -> None:
        self.calls = 0
        dicts, slot_d, slot_k, slot_v = [], [], [], []
        self.tensors = []
        for m in module.modules():
            for d in (m._parameters, m._buffers, m._modules):
                dicts.append(d)
                for k, v in d.items():
                    slot_d.append(d)
                    slot_k.append(k)
                    slot_v.append(v)
            self.tensors.extend(v for v in m._parameters.values() if v is not None)
            self.tensors.extend(v for v in m._buffers.values() if v is not None)
        # parallel tuples, compared with C-level map() calls: the structural part is ~0.09 ms for the 215 M-parameter U-Net
        self.dicts, self.sizes = tuple(dicts), tuple(map(len, dicts))
        self.slots = (tuple(slot_d), tuple(slot_k), tuple(slot_v))

    def _read(self) -> Tuple:
        return tuple((p.data_ptr(), p._version) for p in self.tensors)

    def _structure_changed(self) -> bool:
        if tuple(map(len, self.dicts)) != self.sizes:
            return True
        ds, ks, vs = self.slots
        return not all(map(_operator.is_, map(dict.get, ds, ks), vs))

    def changed(self, module: torch.nn.Module) -> bool:
        self.calls += 1
        if not self.tensors and not self.dicts:
            return True                   # already found stale: the engine is about to be rebuilt with a fresh version object
        stale = self._structure_changed()
        if not stale and self.calls >= _REWALK_EVERY:
            old = self.tensors
            self._walk(module)
            stale = len(old) != len(self.tensors) or any(a is not b for a, b in zip(old, self.tensors))
        if stale:
            self.tensors, self.dicts, self.sizes, self.slots = [], (), (), ((), (), ())      # do not keep dead tensors alive
            return True
        return self._read() != self.value


def _params_version(module: torch.nn.Module) -> "_ParamsVersion":
    return _ParamsVersion(module)


def _fill(arr, values: Sequence[int]) -> None:
    if len(values) > len(arr):
        raise ValueError(f"at most {len(arr)} levels are supported, got {len(values)}")
    for i, v in enumerate(values):
        arr[i] = int(v)


class _Base:
    handle: Optional[int] = None
    _destroy = None

    def __del__(self):
        try:
            if self.handle and self._destroy is not None:
                self._destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _workspace(self, nbytes: int, device: torch.device) -> torch.Tensor:
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            self._ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self._ws


class UNetEngine(_Base):
    """sf_unet_* : one handle per (UNetV0 module weights, dtype, device)."""

    def __init__(self, net: torch.nn.Module, dtype: str):
        lib = _lib.load()
        self.lib = lib
        self._destroy = lib.sf_unet_destroy
        p = next(net.parameters())
        _lib.require_gpu_tensor(p, "UNetV0 parameters")
        self.device = p.device
        hp = net.hparams
        cfg = UnetConfig()
        cfg.n_layers = len(hp["channels"])
        cfg.in_channels = hp["in_channels"]
        for field in ("channels", "factors", "items", "attentions", "cross_attentions", "context_channels"):
            _fill(getattr(cfg, field), hp[field])
        for field in ("attention_heads", "attention_features", "embedding_features", "embedding_max_length",
                      "modulation_features", "resnet_groups"):
            setattr(cfg, field, int(hp[field]))
        cfg.dtype = _lib.DTYPES[dtype]
        cfg.upsample_mode = _lib.UPSAMPLE_MODES[hp.get("upsample_mode", "nearest")]
        cfg.time_fourier_features = int(hp.get("time_fourier_features") or 0)
        cfg.time_no_first_act = 0 if hp.get("time_first_activation", True) else 1
        cfg.attention_out_bias = 1 if hp.get("attention_out_bias", False) else 0
        self.cfg = cfg
        self.hp = dict(hp)
        self.dtype = dtype
        self.version = _params_version(net)
        with torch.cuda.device(self.device):
            table = _lib.TensorTable(((("net." + k), v) for k, v in net.state_dict().items()), self.device)
            h = C.c_void_p()
            check(lib.sf_unet_create(C.byref(cfg), table.array, table.n, _lib.stream_ptr(self.device), C.byref(h)), "sf_unet_create")
            torch.cuda.synchronize(self.device)
        self.handle = h.value
        import os

        if os.environ.get("SF_UNET_BRANCHES"):
            self.set_branches(int(os.environ["SF_UNET_BRANCHES"]))

    def set_branches(self, n: int) -> None:
        """Clip-parallel branches (0 = automatic, 1 = off): see sf_unet_set_branches."""
        check(self.lib.sf_unet_set_branches(self.handle, int(n)), "sf_unet_set_branches")

    def stale(self, net: torch.nn.Module, dtype: str) -> bool:
        return dtype != self.dtype or self.version.changed(net)

    # -- helpers -------------------------------------------------------------------------------
    def _check_inputs(self, x, channels, embedding):
        hp = self.hp
        if x.dim() != 3 or x.shape[1] != hp["in_channels"]:
            raise ValueError(f"expected x of shape (B, {hp['in_channels']}, L), got {tuple(x.shape)}")
        if embedding is None:
            raise AssertionError("ClassifierFreeGuidancePlugin requires embedding")
        B, _, L0 = x.shape
        if tuple(embedding.shape) != (B, hp["embedding_max_length"], hp["embedding_features"]):
            raise ValueError(f"embedding must be (B, {hp['embedding_max_length']}, {hp['embedding_features']}), got {tuple(embedding.shape)}")
        if channels is None or len(channels) != len(hp["channels"]):
            raise AssertionError(f"context `channels` must be a list of {len(hp['channels'])} tensors")
        L = L0
        for d, c in enumerate(channels):
            L //= hp["factors"][d]
            want = (B, hp["context_channels"][d], L)
            if tuple(c.shape) != want:
                raise AssertionError(f"context channels at depth {d}: expected {want}, got {tuple(c.shape)}")

    def _ws_for(self, B: int, L0: int, two: bool, num_steps: int = 0) -> torch.Tensor:
        if num_steps:
            # sized for at least 256 steps: a later call with more steps then finds the same buffer (and the engine its
            # cached step graph, whose kernel nodes hold workspace addresses)
            n = self.lib.sf_vsample_workspace_bytes(self.handle, B, L0, int(two), max(int(num_steps), 256))
        else:
            n = self.lib.sf_unet_workspace_bytes(self.handle, B, L0, int(two))
        if n < 0:
            raise _lib.SyncFusionAmdError(f"unsupported shape B={B}, L0={L0}: {self.lib.sf_last_error().decode()}")
        return self._workspace(n, self.device)

    def forward(self, x: torch.Tensor, sigma: torch.Tensor, channels: Sequence[torch.Tensor], embedding: torch.Tensor,
                embedding_scale: float = 1.0) -> torch.Tensor:
        _lib.require_gpu_tensor(x, "UNetV0.forward")
        self._check_inputs(x, channels, embedding)
        B, _, L0 = x.shape
        with torch.cuda.device(self.device):
            xs = _lib.f32c(x)
            sg = _lib.f32c(sigma).reshape(-1)
            if sg.numel() != B:
                raise ValueError("time/sigma must have one value per batch element")
            ctx = [_lib.f32c(c) for c in channels]
            emb = _lib.f32c(embedding)
            out = torch.empty_like(xs)
            ws = self._ws_for(B, L0, float(embedding_scale) != 1.0)
            check(self.lib.sf_unet_forward(self.handle, xs.data_ptr(), sg.data_ptr(), _lib.ptr_array(ctx), emb.data_ptr(), B, L0,
                                           float(embedding_scale), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                           _lib.stream_ptr(self.device)), "sf_unet_forward")
        return out

    def sample(self, x_noisy: torch.Tensor, num_steps: int, channels: Sequence[torch.Tensor], embedding: torch.Tensor,
               embedding_scale: float = 1.0, use_graph: bool = True) -> torch.Tensor:
        _lib.require_gpu_tensor(x_noisy, "DiffusionModel.sample")
        self._check_inputs(x_noisy, channels, embedding)
        B, _, L0 = x_noisy.shape
        with torch.cuda.device(self.device):
            x = _lib.f32c(x_noisy).clone()  # the caller's noise is not mutated (main/generation.py:69,77-83)
            ctx = [_lib.f32c(c) for c in channels]
            emb = _lib.f32c(embedding)
            ws = self._ws_for(B, L0, float(embedding_scale) != 1.0, int(num_steps))
            check(self.lib.sf_vsample(self.handle, x.data_ptr(), _lib.ptr_array(ctx), emb.data_ptr(), B, L0, int(num_steps),
                                      float(embedding_scale), int(bool(use_graph)), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr(self.device)), "sf_vsample")
        return x

    def launch_count(self) -> int:
        return int(self.lib.sf_unet_launch_count(self.handle))

    def graph_captures(self) -> int:
        return int(self.lib.sf_unet_graph_captures(self.handle))

    def profile_forward(self, x, sigma, channels, embedding, embedding_scale=1.0, with_depth: bool = False):
        """One evaluation with HIP events around every launch -> [(label, ms, algorithmic flops, algorithmic bytes)]
        (``with_depth``: a fifth member, the U-Net depth the launch belongs to, -1 for per-step features)."""
        check(self.lib.sf_unet_profile_enable(self.handle, 1), "sf_unet_profile_enable")
        try:
            self.forward(x, sigma, channels, embedding, embedding_scale)
            recs = []
            for i in range(self.lib.sf_unet_profile_count(self.handle)):
                name = C.create_string_buffer(128)
                ms, fl, by = C.c_float(), C.c_double(), C.c_double()
                check(self.lib.sf_unet_profile_get(self.handle, i, name, 128, C.byref(ms), C.byref(fl), C.byref(by)), "profile_get")
                rec = (name.value.decode(), ms.value, fl.value, by.value)
                recs.append(rec + (self.lib.sf_unet_profile_depth(self.handle, i),) if with_depth else rec)
        finally:
            self.lib.sf_unet_profile_enable(self.handle, 0)
        return recs

    def forward_with_taps(self, x, sigma, channels, embedding, embedding_scale=1.0, cap_floats: int = 1 << 26):
        """tests only: returns (out, {name: (rows, cols) fp32 tensor})."""
        buf = torch.empty(cap_floats, dtype=torch.float32, device=self.device)
        check(self.lib.sf_unet_debug_enable(self.handle, buf.data_ptr(), buf.numel()), "sf_unet_debug_enable")
        try:
            out = self.forward(x, sigma, channels, embedding, embedding_scale)
            torch.cuda.synchronize(self.device)
            taps: Dict[str, torch.Tensor] = {}
            for i in range(self.lib.sf_unet_debug_count(self.handle)):
                name = C.create_string_buffer(128)
                off, rows, cols = C.c_int64(), C.c_int64(), C.c_int32()
                check(self.lib.sf_unet_debug_info(self.handle, i, name, 128, C.byref(off), C.byref(rows), C.byref(cols)), "debug_info")
                taps[name.value.decode()] = buf[off.value: off.value + rows.value * cols.value].reshape(rows.value, cols.value).clone()
        finally:
            self.lib.sf_unet_debug_enable(self.handle, None, 0)
        return out, taps


class EncoderEngine(_Base):
    """sf_encoder1d_*"""

    def __init__(self, enc: torch.nn.Module):
        lib = _lib.load()
        self.lib = lib
        self._destroy = lib.sf_encoder1d_destroy
        p = next(enc.parameters())
        _lib.require_gpu_tensor(p, "Encoder1d parameters")
        self.device = p.device
        hp = enc.hparams
        cfg = EncoderConfig()
        cfg.n_layers = len(hp["factors"])
        cfg.in_channels = hp["in_channels"]
        cfg.channels = hp["channels"]
        _fill(cfg.multipliers, hp["multipliers"])
        _fill(cfg.factors, hp["factors"])
        _fill(cfg.num_blocks, hp["num_blocks"])
        cfg.resnet_groups = hp["resnet_groups"]
        cfg.patch_size = hp["patch_size"]
        self.hp = dict(hp)
        self.version = _params_version(enc)
        with torch.cuda.device(self.device):
            table = _lib.TensorTable(enc.state_dict().items(), self.device)
            h = C.c_void_p()
            check(lib.sf_encoder1d_create(C.byref(cfg), table.array, table.n, _lib.stream_ptr(self.device), C.byref(h)), "sf_encoder1d_create")
            torch.cuda.synchronize(self.device)
        self.handle = h.value

    def stale(self, enc: torch.nn.Module) -> bool:
        return self.version.changed(enc)

    def forward(self, y: torch.Tensor) -> List[torch.Tensor]:
        _lib.require_gpu_tensor(y, "Encoder1d.forward")
        hp = self.hp
        if y.dim() != 3 or y.shape[1] != hp["in_channels"]:
            raise ValueError(f"expected (B, {hp['in_channels']}, L), got {tuple(y.shape)}")
        B, _, L0 = y.shape
        with torch.cuda.device(self.device):
            ys = _lib.f32c(y)
            outs = [torch.empty(B, hp["channels"] * hp["multipliers"][0], L0, dtype=torch.float32, device=self.device)]
            L = L0
            for i, f in enumerate(hp["factors"]):
                L = (L - 1) // f + 1
                outs.append(torch.empty(B, hp["channels"] * hp["multipliers"][i + 1], L, dtype=torch.float32, device=self.device))
            n = self.lib.sf_encoder1d_workspace_bytes(self.handle, B, L0)
            if n < 0:
                raise _lib.SyncFusionAmdError(self.lib.sf_last_error().decode())
            ws = self._workspace(n, self.device)
            check(self.lib.sf_encoder1d_forward(self.handle, ys.data_ptr(), B, L0, _lib.ptr_array(outs), ws.data_ptr(), ws.numel(),
                                                _lib.stream_ptr(self.device)), "sf_encoder1d_forward")
        return outs


class OnsetNetEngine(_Base):
    """sf_onsetnet_*"""

    def __init__(self, net: torch.nn.Module, dtype: str):
        lib = _lib.load()
        self.lib = lib
        self._destroy = lib.sf_onsetnet_destroy
        p = next(net.parameters())
        _lib.require_gpu_tensor(p, "VideoOnsetNet parameters")
        self.device = p.device
        self.dtype = dtype
        self.version = _params_version(net)
        with torch.cuda.device(self.device):
            sd = {k: v for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
            table = _lib.TensorTable(sd.items(), self.device)
            h = C.c_void_p()
            check(lib.sf_onsetnet_create(table.array, table.n, _lib.DTYPES[dtype], _lib.stream_ptr(self.device), C.byref(h)),
                  "sf_onsetnet_create")
            torch.cuda.synchronize(self.device)
        self.handle = h.value

    def stale(self, net: torch.nn.Module) -> bool:
        return self.version.changed(net) or getattr(net, "compute_dtype", self.dtype) != self.dtype

    def forward(self, x: torch.Tensor, taps: Optional[dict] = None, cap_floats: int = 1 << 27) -> torch.Tensor:
        N, _, T, H, W = x.shape
        with torch.cuda.device(self.device):
            xs = _lib.f32c(x)
            out = torch.empty(N, T, dtype=torch.float32, device=self.device)
            n = self.lib.sf_onsetnet_workspace_bytes(self.handle, N, T, H, W)
            if n < 0:
                raise _lib.SyncFusionAmdError(self.lib.sf_last_error().decode())
            ws = self._workspace(n, self.device)
            buf = None
            if taps is not None:
                buf = torch.empty(cap_floats, dtype=torch.float32, device=self.device)
                check(self.lib.sf_onsetnet_debug_enable(self.handle, buf.data_ptr(), buf.numel()), "debug_enable")
            try:
                check(self.lib.sf_onsetnet_forward(self.handle, xs.data_ptr(), N, T, H, W, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   _lib.stream_ptr(self.device)), "sf_onsetnet_forward")
                if taps is not None:
                    torch.cuda.synchronize(self.device)
                    for i in range(self.lib.sf_onsetnet_debug_count(self.handle)):
                        name = C.create_string_buffer(128)
                        off, rows, cols = C.c_int64(), C.c_int64(), C.c_int32()
                        check(self.lib.sf_onsetnet_debug_info(self.handle, i, name, 128, C.byref(off), C.byref(rows), C.byref(cols)), "debug_info")
                        taps[name.value.decode()] = buf[off.value: off.value + rows.value * cols.value].reshape(rows.value, cols.value).clone()
            finally:
                if taps is not None:
                    self.lib.sf_onsetnet_debug_enable(self.handle, None, 0)
        return out
