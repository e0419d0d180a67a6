"""Device resampler with the call shape of ``torchaudio.functional.resample``.

The reference resamples every generated clip on the CPU (main/generation.py:91-98:
``torchaudio.functional.resample(gen[i, :, :cut_length].cpu(), orig_freq=sample_rate, new_freq=downsample_rate)``);
this keeps the clip on the GPU (``sf_resampler_forward``: windowed-sinc polyphase filter, torchaudio 0.13.1 defaults).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib

_cache: Dict[Tuple[int, int, int, float], int] = {}


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """``(..., time)`` fp32 tensor on the GPU -> ``(..., ceil(new_freq * time / orig_freq))``."""
    _lib.require_gpu_tensor(waveform, "resample")
    if int(orig_freq) == int(new_freq):
        return waveform
    lib = _lib.load()
    key = (int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff))
    if key not in _cache:
        h = C.c_void_p()
        with torch.cuda.device(waveform.device):
            _lib.check(lib.sf_resampler_create(key[0], key[1], key[2], key[3], C.byref(h)), "sf_resampler_create")
        _cache[key] = h.value
    h = _cache[key]
    x = _lib.f32c(waveform)
    L = x.shape[-1]
    R = x.numel() // L
    Lout = lib.sf_resampler_out_length(h, L)
    out = torch.empty(x.shape[:-1] + (Lout,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.sf_resampler_forward(h, x.data_ptr(), R, L, out.data_ptr(), _lib.stream_ptr(x.device)), "sf_resampler_forward")
    return out
