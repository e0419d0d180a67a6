"""Oracle: ``torchaudio.functional.resample`` (torchaudio==0.13.1, requirements.txt:19) restated on the CPU.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED: torchaudio is not installed in this image and
is not part of /root/reference; this follows the published algorithm of ``_get_sinc_resample_kernel`` /
``_apply_sinc_resample_kernel`` (windowed sinc, Hann window, lowpass_filter_width=6, rolloff=0.99), anchored on the
reference's call site main/generation.py:91-98 and on analytic properties (tests/test_oracle_cpu.py).
"""
from __future__ import annotations

import math

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    gcd = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // gcd, int(new_freq) // gcd
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1)[:, None, None] / new + idx        # int64 / int -> float32, then + float64 (as upstream)
    t = t * base_freq
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=t.dtype), t.sin() / t)
    kernels = kernels * (window * scale)
    return kernels.to(torch.float32), width, orig, new


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    if int(orig_freq) == int(new_freq):
        return waveform
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    shape = waveform.shape
    wav = waveform.reshape(-1, shape[-1]).to(torch.float32)
    num, length = wav.shape
    wav = torch.nn.functional.pad(wav, (width, width + orig))
    res = torch.nn.functional.conv1d(wav[:, None], kernel, stride=orig)
    res = res.transpose(1, 2).reshape(num, -1)
    target = int(math.ceil(new * length / orig))
    return res[..., :target].reshape(shape[:-1] + (target,))
