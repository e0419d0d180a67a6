"""Frame-directory side of the onset path: what ``main/dataset_onset.py`` builds for ``VideoOnsetNet`` --
the per-video chunk table (:62-116) and the frames of one chunk (:121-165) -- with the per-pixel work (ToTensor -> antialiased
Resize -> Normalize -> (C, T, H, W)) in the HIP library (``frames_to_clip`` / ``sf_frames_preprocess``).

Layout the reference reads (Greatest Hits as it preprocesses it):
    <root>/<sample>/<sample>.metadata.json      {"processed": {"video_frame_rate": f, "video_duration": d}}
    <root>/<sample>/<sample>.times.csv          ``time,label`` lines (no header)
    <root>/<sample>/frames/*.jpg                one image per frame, naturally sorted
Image decoding is host I/O through PIL (as in the reference); everything after the uint8 pixels runs on the device.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch

from .input_pipeline import frame_labels, frames_to_clip

Tensor = torch.Tensor


def natural_sorted(items: Sequence[str]) -> List[str]:
    """natsort's default order for file names: digit runs compare as numbers (1, 2, 10 instead of 1, 10, 2)."""
    def key(s: str):
        return [(0, int(t), "") if t.isdigit() else (1, 0, t) for t in re.split(r"(\d+)", s) if t != ""]

    return sorted(items, key=key)


def chunk_table(root_dir: str, samples: Sequence[str], chunk_length_in_seconds: float = 2.0, annotations_file_suffix: str = ".times.csv",
                metadata_file_suffix: str = ".metadata.json") -> List[dict]:
    """main/dataset_onset.py:62-116: whole chunks of ``chunk_length_in_seconds`` per video with their frame ranges and 0/1 labels."""
    out: List[dict] = []
    for sample in samples:
        with open(os.path.join(root_dir, sample, f"{sample}{metadata_file_suffix}"), "r") as f:
            meta = json.load(f)
        frame_rate = meta["processed"]["video_frame_rate"]
        duration = meta["processed"]["video_duration"]
        num_chunks = int(duration / chunk_length_in_seconds)
        times: List[float] = []
        with open(os.path.join(root_dir, sample, f"{sample}{annotations_file_suffix}"), "r") as f:
            for line in f.read().splitlines():
                if line.strip():
                    times.append(float(line.split(",")[0]))
        for i in range(num_chunks):
            t0 = i * chunk_length_in_seconds
            t1 = t0 + chunk_length_in_seconds
            out.append(dict(video_name=sample, frames_path=os.path.join(root_dir, sample, "frames"), start_time=t0, end_time=t1,
                            start_frame=int(t0 * frame_rate), end_frame=int(t1 * frame_rate), frame_rate=frame_rate,
                            labels=frame_labels(times, t0, chunk_length_in_seconds, frame_rate)))
    return out


def read_frames_u8(paths: Sequence[str], pin: bool = True) -> Tensor:
    """Decode images (PIL, ``convert('RGB')`` as main/dataset_onset.py:155) into one ``(T, H, W, 3)`` uint8 tensor (pinned for the upload)."""
    from PIL import Image
    import numpy as np

    arrs = [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    if not arrs:
        raise ValueError("no frames")
    if any(a.shape != arrs[0].shape for a in arrs):
        raise ValueError("frames of one chunk differ in size")
    t = torch.from_numpy(np.stack(arrs))
    return t.pin_memory() if pin and torch.cuda.is_available() else t


def chunk_clip(chunk: dict, device, frame_file_suffix: str = ".jpg", size: Tuple[int, int] = (112, 112)) -> Tensor:
    """Frames of one chunk -> ``(3, T, 112, 112)`` float32 on ``device``: main/dataset_onset.py:121-165 (``__getitem__`` +
    ``read_image_and_apply_transforms``) with the transform chain in one HIP pass."""
    frames = natural_sorted(glob.glob(f"{chunk['frames_path']}/*{frame_file_suffix}"))[chunk["start_frame"]: chunk["end_frame"]]
    u8 = read_frames_u8(frames).to(device, non_blocking=True)
    return frames_to_clip(u8[None], size)[0]


def iter_clips(chunks: Sequence[dict], batch_size: int, device, frame_file_suffix: str = ".jpg") -> Iterator[Tuple[Tensor, Tensor, List[dict]]]:
    """Batches ``(frames (N, 3, T, 112, 112), labels (N, T), chunk dicts)`` as the reference's DataLoader stacks them."""
    for i in range(0, len(chunks), batch_size):
        part = list(chunks[i: i + batch_size])
        clips = torch.stack([chunk_clip(c, device, frame_file_suffix) for c in part])
        yield clips, torch.stack([c["labels"] for c in part]).to(device), part
