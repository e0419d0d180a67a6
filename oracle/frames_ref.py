"""Oracle: the frame transform of the reference's onset dataset -- TEST INFRASTRUCTURE (see oracle/__init__.py).

main/dataset_onset.py:47-50,152-165: ``ToTensor`` (uint8 HWC -> float CHW / 255), ``Resize((112, 112), antialias=True)``,
``Normalize(mean, std)``, then ``(T, C, H, W) -> (C, T, H, W)``.  torchvision is absent here; for tensors its ``Resize`` is
``torch.nn.functional.interpolate(mode="bilinear", antialias=True, align_corners=False)``, which IS present, so this oracle
is pinned to the same ATen kernel the reference runs.  ``aa_weights`` restates that kernel's separable triangle filter
(ATen UpSampleKernel `_compute_indices_min_size_weights_aa`) in numpy; tests check the restatement against ``interpolate``."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)   # main/dataset_onset.py:49
STD = (0.229, 0.224, 0.225)


def frames_transform(frames_u8: torch.Tensor, size=(112, 112), mean=MEAN, std=STD) -> torch.Tensor:
    """frames_u8: (N, T, H, W, 3) uint8 -> (N, 3, T, size[0], size[1]) float32."""
    N, T, H, W, C = frames_u8.shape
    x = frames_u8.reshape(N * T, H, W, C).permute(0, 3, 1, 2).to(torch.float32) / 255.0           # ToTensor
    x = F.interpolate(x, size=size, mode="bilinear", antialias=True, align_corners=False)           # Resize(antialias=True)
    m = torch.tensor(mean, dtype=torch.float32).reshape(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).reshape(1, 3, 1, 1)
    x = (x - m) / s                                                                                   # Normalize
    return x.reshape(N, T, C, size[0], size[1]).permute(0, 2, 1, 3, 4).contiguous()


def aa_weights(in_size: int, out_size: int):
    """Per output index: (first input index, weights) of the antialiased bilinear (triangle) filter, align_corners=False."""
    scale = in_size / out_size
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    out = []
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((j + xmin - center + 0.5) * invscale)) for j in range(xsize)], dtype=np.float64)
        w = w / w.sum()
        out.append((xmin, w.astype(np.float32)))
    return out


def resize_aa_numpy(x: np.ndarray, size) -> np.ndarray:
    """(..., H, W) float32 -> (..., size[0], size[1]) with the restated filter (horizontal pass, then vertical, as ATen does)."""
    H, W = x.shape[-2:]
    wy, wx = aa_weights(H, size[0]), aa_weights(W, size[1])
    tmp = np.zeros(x.shape[:-1] + (size[1],), dtype=np.float32)
    for j, (x0, w) in enumerate(wx):
        tmp[..., j] = (x[..., x0:x0 + len(w)] * w).sum(-1)
    out = np.zeros(x.shape[:-2] + (size[0], size[1]), dtype=np.float32)
    for i, (y0, w) in enumerate(wy):
        out[..., i, :] = (tmp[..., y0:y0 + len(w), :] * w[:, None]).sum(-2)
    return out
