"""Input side of the generation path: the tensor contract the reference's datasets produce,
built on the device where it is arithmetic and kept as plain host logic where it is bookkeeping.

* ``frames_to_clip``    decoded uint8 frames -> ``(N, 3, T, 112, 112)`` normalised clips: ToTensor -> Resize(antialias) ->
                        Normalize -> (C, T, H, W) of main/dataset_onset.py:47-50,152-165 in one HIP pass (``sf_frames_preprocess``)
* ``times_to_track``    onset times -> one-hot impulse track, ``onset[:, int(t * sr)] = 1.0`` (main/dataset_diffusion.py:58-72),
                        HIP (``sf_times_to_track``)
* ``frame_labels``      onset times -> per-frame 0/1 labels of a chunk (main/dataset_onset.py:88-101)
* ``slice_chunks``      the chunking generator ``_get_slices`` (main/dataset_diffusion.py:47-108): fixed-size chunks, skip chunks
                        without an onset in the first ``onset_check_length`` samples, ``cut_prefix`` zeroing, conditioning chunk
* ``collate_fn``        main/dataset_diffusion.py:134-143 (stack, right-pad the conditioning chunks)
* ``PinnedPrefetcher``  double-buffered pinned-memory upload on a side stream, so that batch i+1 crosses PCIe while batch i
                        is being sampled (the reference relies on DataLoader(pin_memory=True), main/datamodule_diffusion.py:36-44)

The tar shards themselves are read by ``syncfusion_amd/shards.py`` (standard library; PCM wav + the times csv files, source audio
resampled by the HIP resampler).  JPEG / video decoding stays out of scope (host I/O through third-party codecs).
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

from . import _lib

Tensor = torch.Tensor
IMAGENET_MEAN = (0.485, 0.456, 0.406)   # main/dataset_onset.py:49
IMAGENET_STD = (0.229, 0.224, 0.225)


def frames_to_clip(frames_u8: Tensor, size: Tuple[int, int] = (112, 112), mean: Sequence[float] = IMAGENET_MEAN,
                   std: Sequence[float] = IMAGENET_STD) -> Tensor:
    """``(N, T, H, W, 3)`` uint8 (device) -> ``(N, 3, T, size[0], size[1])`` float32: the input of ``VideoOnsetNet``."""
    import ctypes as C

    _lib.require_gpu_tensor(frames_u8, "frames_to_clip")
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 5 or frames_u8.shape[-1] != 3:
        raise ValueError(f"expected uint8 frames of shape (N, T, H, W, 3), got {frames_u8.dtype} {tuple(frames_u8.shape)}")
    lib = _lib.load()
    N, T, H, W, _ = frames_u8.shape
    fr = frames_u8.contiguous()
    out = torch.empty(N, 3, T, size[0], size[1], dtype=torch.float32, device=fr.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(fr.device):
        _lib.check(lib.sf_frames_preprocess(fr.data_ptr(), N, T, H, W, int(size[0]), int(size[1]), m, s, out.data_ptr(), _lib.stream_ptr(fr.device)),
                   "sf_frames_preprocess")
    return out


def times_to_track(times: Sequence[Sequence[float]], sample_rate: float, length: int, device) -> Tensor:
    """Per-clip onset times (seconds) -> ``(B, 1, length)`` one-hot track on ``device`` (indices ``int(t * sample_rate)``;
    times that fall outside the chunk are dropped, as slicing the full-length track does in the reference)."""
    device = torch.device(device)
    _lib.require_gpu_tensor(torch.empty(0, device=device), "times_to_track")
    lib = _lib.load()
    B = len(times)
    flat = [float(t) for ts in times for t in ts]
    clip = [b for b, ts in enumerate(times) for _ in ts]
    track = torch.empty(B, 1, length, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        tt = torch.tensor(flat, dtype=torch.float64).to(device) if flat else None
        cc = torch.tensor(clip, dtype=torch.int32).to(device) if flat else None
        _lib.check(lib.sf_times_to_track(tt.data_ptr() if flat else None, cc.data_ptr() if flat else None, len(flat), float(sample_rate), B, int(length),
                                         track.data_ptr(), _lib.stream_ptr(device)), "sf_times_to_track")
    return track


def frame_labels(onset_times: Sequence[float], chunk_start_time: float, chunk_length_in_seconds: float, frame_rate: float) -> Tensor:
    """Per-frame onset labels of one chunk (main/dataset_onset.py:88-101)."""
    n = int(chunk_length_in_seconds * frame_rate)
    labels = torch.zeros(n)
    end = chunk_start_time + chunk_length_in_seconds
    for t in onset_times:
        if chunk_start_time <= t < end:
            labels[int((t - chunk_start_time) * frame_rate)] = 1
    return labels


def slice_chunks(wav: Tensor, sr: int, onset_times: Dict[float, Optional[str]], pred_onset_times: Optional[Dict[float, Optional[str]]], filename: str,
                 chunk_size: int, onset_check_length: Optional[int] = None, shift_augment: bool = False, cut_prefix: bool = True,
                 one_chunk_per_track: bool = False, rng: Optional[random.Random] = None) -> Iterator[Tuple[Tensor, Tensor, Tensor, str, str]]:
    """One sample of the reference's pipeline -> ``(wav_chunk, pred_onset_chunk, cond_chunk, text, filename)`` tuples
    (main/dataset_diffusion.py:47-108; ``rng`` replaces the module-level ``random`` for reproducible tests)."""
    rng = rng or random
    onset_check_length = onset_check_length or chunk_size
    channels, length = wav.shape
    if pred_onset_times is None:
        pred_onset_times = onset_times
    onset_idx = [int(k * sr) for k in onset_times.keys()]
    texts = [t for t in onset_times.values() if t is not None and "None" not in t]
    assert onset_idx
    text = rng.choice(texts) if texts else ""
    onset = torch.zeros_like(wav)
    onset[:, onset_idx] = 1.0
    pred_idx = [int(k * sr) for k in pred_onset_times.keys()]
    assert pred_idx
    pred_onset = torch.zeros_like(wav)
    pred_onset[:, pred_idx] = 1.0
    assert length >= chunk_size
    shift = 0
    if shift_augment:
        max_shift = length - (length // chunk_size) * chunk_size
        shift = rng.randint(0, max_shift)
    done = False
    for i in range(length // chunk_size):
        if done and one_chunk_per_track:
            break
        start = min(length - chunk_size, i * chunk_size + shift)
        end = start + chunk_size
        wav_chunk = wav[:, start:end].clone()
        onset_chunk = onset[:, start:end]
        pred_chunk = pred_onset[:, start:end]
        if torch.all(onset_chunk[:, :onset_check_length] == 0.0):
            if one_chunk_per_track:
                break
            continue
        onset_indices = torch.nonzero(onset_chunk[0]).squeeze(-1)
        if cut_prefix:
            wav_chunk[:, : onset_indices[0]] = 0.0
        k = rng.randint(0, len(onset_indices) - 1)                      # _get_cond_chunk (:36-44)
        s0 = int(onset_indices[k])
        s1 = wav_chunk.shape[1] if k == len(onset_indices) - 1 else int(onset_indices[k + 1])
        done = True
        yield wav_chunk, pred_chunk, wav_chunk[:, s0:s1], text, filename


def collate_fn(data):
    """main/dataset_diffusion.py:134-143."""
    waveforms, onset_tensors, cond_chunks, texts, filenames = zip(*data)
    max_length = max(c.size(1) for c in cond_chunks)
    cond = torch.stack([torch.nn.functional.pad(c, (0, max_length - c.size(1))) for c in cond_chunks], dim=0)
    return torch.stack(waveforms, dim=0), torch.stack(onset_tensors, dim=0), cond, texts, filenames


class PinnedPrefetcher:
    """Iterate ``batches`` (tuples whose tensor members live on the host) with the tensors of batch i+1 already in flight
    to ``device`` on a side stream while the caller works on batch i: two pinned staging slots per tensor position, one
    ``torch.cuda.Event`` per slot to fence reuse; non-tensor members pass through."""

    def __init__(self, batches: Iterable, device, depth: int = 2):
        self.it = iter(batches)
        self.device = torch.device(device)
        _lib.require_gpu_tensor(torch.empty(0, device=self.device), "PinnedPrefetcher")
        self.depth = max(2, int(depth))
        self.stream = torch.cuda.Stream(self.device)
        self.slots: List[Dict[int, Tensor]] = [dict() for _ in range(self.depth)]
        self.done: List[Optional[torch.cuda.Event]] = [None] * self.depth
        self.n = 0
        self.queue: List[Tuple[tuple, torch.cuda.Event]] = []

    def _stage(self, batch):
        slot = self.n % self.depth
        self.n += 1
        if self.done[slot] is not None:
            self.done[slot].synchronize()       # the copy that last used this pinned slot has finished
        out = []
        with torch.cuda.stream(self.stream):
            for i, v in enumerate(batch):
                if isinstance(v, torch.Tensor) and not v.is_cuda:
                    pin = self.slots[slot].get(i)
                    if pin is None or pin.shape != v.shape or pin.dtype != v.dtype:
                        pin = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                        self.slots[slot][i] = pin
                    pin.copy_(v)
                    out.append(pin.to(self.device, non_blocking=True))
                else:
                    out.append(v)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.done[slot] = ev
        return tuple(out), ev

    def __iter__(self):
        return self

    def __next__(self):
        while len(self.queue) < self.depth - 1 + 1:
            try:
                self.queue.append(self._stage(next(self.it)))
            except StopIteration:
                break
        if not self.queue:
            raise StopIteration
        batch, ev = self.queue.pop(0)
        torch.cuda.current_stream(self.device).wait_event(ev)   # the consumer's stream sees the upload; the host does not block
        for v in batch:
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(torch.cuda.current_stream(self.device))
        return batch
