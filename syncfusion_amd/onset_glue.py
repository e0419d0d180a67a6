"""In-memory version of the reference's file-mediated onset hand-off (SURVEY.md sections 3.5, 8a-7).

Reference chain: ``log_annotations`` thresholds RAW logits at 0.5, converts frame index to seconds
``(idx + start_frame) / frame_rate`` and writes them with ``"%.4f"`` (main/module_onset.py:160-183);
``_get_slices`` later turns each time ``t`` into a one-hot sample ``track[:, int(t * sr)] = 1``
(main/dataset_diffusion.py:69-72).  ``sf_onsets_to_track`` does the same on the device, including the
4-decimal rounding.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


@torch.no_grad()
def onsets_to_track(logits: torch.Tensor, length: int, frame_rate: float = 15.0, sample_rate: float = 48000.0,
                    threshold: float = 0.5, start_frame: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(N, T) raw logits -> (N, 1, length) one-hot impulse track (fp32) on the same device."""
    _lib.require_gpu_tensor(logits, "onsets_to_track")
    lib = _lib.load()
    lg = _lib.f32c(logits)
    N, T = lg.shape
    track = torch.empty(N, 1, length, dtype=torch.float32, device=lg.device)
    sf = None
    if start_frame is not None:
        sf = start_frame.to(device=lg.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(lg.device):
        _lib.check(lib.sf_onsets_to_track(lg.data_ptr(), N, T, sf.data_ptr() if sf is not None else None, float(frame_rate),
                                          float(sample_rate), float(threshold), track.data_ptr(), int(length),
                                          _lib.stream_ptr(lg.device)), "sf_onsets_to_track")
    return track


@torch.no_grad()
def cut_prefix_crop(gen: torch.Tensor, y: torch.Tensor, cut_length: Optional[int] = None) -> torch.Tensor:
    """``gen[i, :, :first_onset(y[i])] = 0`` then crop to ``cut_length`` (main/generation.py:86-89,100) in one device
    pass for the whole batch; raises ``IndexError`` like the reference when a track has no onset."""
    _lib.require_gpu_tensor(gen, "cut_prefix_crop")
    lib = _lib.load()
    g = _lib.f32c(gen)
    B, C, L = g.shape
    yt = _lib.f32c(y.to(g.device)).reshape(B, -1)
    if yt.shape[1] < L:
        raise ValueError("onset track shorter than the generated audio")
    yt = yt[:, :L].contiguous() if yt.shape[1] != L else yt
    Lc = int(cut_length or L)
    out = torch.empty(B, C, Lc, dtype=torch.float32, device=g.device)
    first = torch.empty(B, dtype=torch.int32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.sf_cut_prefix_crop(g.data_ptr(), yt.data_ptr(), B, C, L, Lc, out.data_ptr(), first.data_ptr(),
                                          _lib.stream_ptr(g.device)), "sf_cut_prefix_crop")
    empty = (first >= L).nonzero()
    if empty.numel():
        raise IndexError(f"clip {int(empty[0])}: cut_prefix=True needs at least one onset in y")
    return out
