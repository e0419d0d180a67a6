"""ctypes binding of ``libsyncfusion_amd.so`` (the C ABI declared in include/syncfusion_amd.h).

The product path has no CPU fallback: if the shared library is missing, or a CPU tensor reaches a
forward/sample call, this module raises.  ``torch`` is used for device memory and streams only;
no torch type crosses the ABI (raw ``data_ptr()`` integers and the HIP stream handle do).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

SF_MAX_DEPTH = 12
SF_F32, SF_BF16, SF_F16, SF_F32X = 0, 1, 2, 3
UPSAMPLE_MODES = {"nearest": 0, "transpose": 1}
DTYPES = {"fp32": SF_F32, "float32": SF_F32, "f32": SF_F32, "bf16": SF_BF16, "bfloat16": SF_BF16, "fp16": SF_F16, "float16": SF_F16, "f16": SF_F16, "half": SF_F16,
          "fp32x": SF_F32X}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SF_LIB_PATH") or os.path.join(_HERE, "lib", "libsyncfusion_amd.so")   # SF_LIB_PATH: A/B builds of the HIP library (tuning aid)


class SfTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class UnetConfig(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("in_channels", C.c_int32),
        ("channels", C.c_int32 * SF_MAX_DEPTH), ("factors", C.c_int32 * SF_MAX_DEPTH),
        ("items", C.c_int32 * SF_MAX_DEPTH), ("attentions", C.c_int32 * SF_MAX_DEPTH),
        ("cross_attentions", C.c_int32 * SF_MAX_DEPTH), ("context_channels", C.c_int32 * SF_MAX_DEPTH),
        ("attention_heads", C.c_int32), ("attention_features", C.c_int32),
        ("embedding_features", C.c_int32), ("embedding_max_length", C.c_int32),
        ("modulation_features", C.c_int32), ("resnet_groups", C.c_int32), ("dtype", C.c_int32),
        ("upsample_mode", C.c_int32),
        ("time_fourier_features", C.c_int32), ("time_no_first_act", C.c_int32), ("attention_out_bias", C.c_int32),
    ]


class EncoderConfig(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("in_channels", C.c_int32), ("channels", C.c_int32),
        ("multipliers", C.c_int32 * (SF_MAX_DEPTH + 1)), ("factors", C.c_int32 * SF_MAX_DEPTH),
        ("num_blocks", C.c_int32 * SF_MAX_DEPTH), ("resnet_groups", C.c_int32), ("patch_size", C.c_int32),
    ]


# every symbol include/syncfusion_amd.h declares: (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "sf_version": (C.c_char_p, []),
    "sf_last_error": (C.c_char_p, []),
    "sf_device_ok": (_I, []),
    "sf_clock_probe_start": (_I, [C.c_double, _P]),
    "sf_clock_probe_read": (_I, [C.POINTER(C.c_double)]),
    "sf_unet_create": (_I, [C.POINTER(UnetConfig), C.POINTER(SfTensor), _I, _P, C.POINTER(_P)]),
    "sf_unet_destroy": (None, [_P]),
    "sf_unet_param_count": (_I, [C.POINTER(UnetConfig)]),
    "sf_unet_param_name": (_I, [C.POINTER(UnetConfig), _I, C.c_char_p, _I, C.POINTER(_L)]),
    "sf_unet_workspace_bytes": (_L, [_P, _I, _I, _I]),
    "sf_vsample_workspace_bytes": (_L, [_P, _I, _I, _I, _I]),
    "sf_unet_forward": (_I, [_P, _P, _P, C.POINTER(_P), _P, _I, _I, _F, _P, _P, _L, _P]),
    "sf_vsample": (_I, [_P, _P, C.POINTER(_P), _P, _I, _I, _I, _F, _I, _P, _L, _P]),
    "sf_unet_debug_enable": (_I, [_P, _P, _L]),
    "sf_unet_debug_count": (_I, [_P]),
    "sf_unet_debug_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(C.c_int32)]),
    "sf_unet_launch_count": (_I, [_P]),
    "sf_unet_graph_captures": (_I, [_P]),
    "sf_unet_set_branches": (_I, [_P, _I]),
    "sf_unet_profile_enable": (_I, [_P, _I]),
    "sf_unet_profile_count": (_I, [_P]),
    "sf_unet_profile_get": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sf_unet_profile_depth": (_I, [_P, _I]),
    "sf_encoder1d_create": (_I, [C.POINTER(EncoderConfig), C.POINTER(SfTensor), _I, _P, C.POINTER(_P)]),
    "sf_encoder1d_destroy": (None, [_P]),
    "sf_encoder1d_workspace_bytes": (_L, [_P, _I, _I]),
    "sf_encoder1d_forward": (_I, [_P, _P, _I, _I, C.POINTER(_P), _P, _L, _P]),
    "sf_onsetnet_create": (_I, [C.POINTER(SfTensor), _I, _I, _P, C.POINTER(_P)]),
    "sf_onsetnet_destroy": (None, [_P]),
    "sf_onsetnet_workspace_bytes": (_L, [_P, _I, _I, _I, _I]),
    "sf_onsetnet_forward": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _L, _P]),
    "sf_onsetnet_debug_enable": (_I, [_P, _P, _L]),
    "sf_onsetnet_debug_count": (_I, [_P]),
    "sf_onsetnet_debug_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(C.c_int32)]),
    "sf_onsets_to_track": (_I, [_P, _I, _I, _P, _F, _F, _F, _P, _I, _P]),
    "sf_frames_preprocess": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sf_times_to_track": (_I, [_P, _P, _I, C.c_double, _I, _I, _P, _P]),
    "sf_cut_prefix_crop": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sf_resampler_create": (_I, [_I, _I, _I, _F, C.POINTER(_P)]),
    "sf_resampler_destroy": (None, [_P]),
    "sf_resampler_out_length": (_I, [_P, _I]),
    "sf_resampler_forward": (_I, [_P, _P, _I, _I, _P, _P]),
    "sf_op_conv1d_cl": (_I, [_I, _P, _P, _P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P]),
    "sf_op_conv1d_bwd_workspace_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "sf_op_conv1d_bwd_cl": (_I, [_P, _P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "sf_op_conv1d_bwd_cl_act": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "sf_op_conv1d_bwd_cl_x": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "sf_op_conv1d_dgrad_pack_bytes": (_L, [_I, _I, _I]),
    # (dtype, x, w, bias, gamma, beta, groups, eps, residual, B, L, C, N, taps, pad, out, dgrad_pack, dgrad_pack_bytes, ws, ws_bytes, stream)
    "sf_op_conv1d_train_fwd": (_I, [_I, _P, _P, _P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P, _L, _P]),
    # (dtype, x, act, stats, w, dgrad_pack, gamma, beta, groups, eps, dy, dx_add, B, L, C, N, taps, pad, dx, dw, db, dgb, ws, ws_bytes, stream)
    "sf_op_conv1d_bwd_cl_p": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "sf_op_conv1d_train_images": (_I, [_I, _P, _I, _I, _I, _I, _I, _I, _I]),
    "sf_train_pack_many": (_I, [_P, _I, _I, _P]),
    # (dtype, x, w, fw, fwx, bias, gamma, beta, groups, eps, residual, B, L, C, N, taps, pad, out, ws, ws_bytes, stream)
    "sf_op_conv1d_train_fwd_pk": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P, _P, _L, _P]),
    "sf_op_ln_modulate_bwd_add": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _L, _P]),
    "sf_op_gn_silu_train_stats_floats": (_L, [_I, _I, _I, _I]),
    "sf_op_gn_silu_train": (_I, [_P, _P, _P, _I, _F, _I, _I, _I, _P, _P, _P]),
    "sf_op_ln_modulate_bwd_workspace_bytes": (_L, [_I, _I, _I]),
    "sf_op_ln_modulate_bwd": (_I, [_P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _L, _P]),
    "sf_op_attention_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _L, _P]),
    "sf_op_attention_fwd_lse": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sf_op_attention_fwd_lse_x": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sf_op_attention_bwd_lse": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _L, _P]),
    "sf_op_attention_bwd_lse_x": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _L, _P]),
    "sf_op_length_sums_workspace_bytes": (_L, [_I, _I, _I]),
    "sf_op_length_sums": (_I, [_P, _P, _I, _I, _I, _P, _P, _L, _P]),
    "sf_bench_conv1d": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float)]),
    "sf_op_gn_silu": (_I, [_I, _P, _P, _P, _I, _F, _I, _I, _I, _P, _P, _L, _P]),
    "sf_op_inject_prenorm_proj_workspace_bytes": (_L, [_I, _I, _I, _I, _I]),
    "sf_op_inject_prenorm_proj": (_I, [_I, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "sf_op_resnet_mod_cb_workspace_bytes": (_L, [_I, _I, _I]),
    "sf_op_resnet_mod_cb": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _F, _I, _I, _I, _I, _P, _P, _P, _L, _P]),
    "sf_bench_conv_cb": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float)]),
    "sf_op_ln_modulate": (_I, [_I, _P, _P, _F, _I, _I, _I, _P, _P]),
    "sf_op_attention": (_I, [_I, _P, _P, _I, _I, _I, _I, _P, _P]),
}

_lib: Optional[C.CDLL] = None


class SyncFusionAmdError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once) and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SyncFusionAmdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C syncfusion_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sf_last_error()
        raise SyncFusionAmdError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def require_gpu_tensor(t: torch.Tensor, where: str) -> None:
    if not t.is_cuda:
        raise SyncFusionAmdError(f"{where}: expected a tensor on an MI355X (cuda/hip) device, got {t.device}; "
                                 "this build has no CPU execution path")


def stream_ptr(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class TensorTable:
    """Keeps the (name, fp32 contiguous device tensor) pairs alive while the C side reads them."""

    def __init__(self, named: Iterable[Tuple[str, torch.Tensor]], device: torch.device):
        self.keep: List[torch.Tensor] = []
        self.names: List[bytes] = []
        items = list(named)
        self.array = (SfTensor * max(1, len(items)))()
        self.n = len(items)
        for i, (name, t) in enumerate(items):
            tt = f32c(t).to(device)
            self.keep.append(tt)
            self.names.append(name.encode())
            self.array[i].name = self.names[-1]
            self.array[i].data = tt.data_ptr()
            self.array[i].numel = tt.numel()


def ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
