"""Shard reader of the generation / training input side: the WebDataset tar layout the reference's
``create_sfx_dataset`` consumes (main/dataset_diffusion.py:111-131), read with the standard library only.

A shard is a tar archive whose members ``<key>.<ext>`` are grouped by key; the reference uses three members per sample:

    <key>.resampled.wav     the sound effect (decoded to float32 in [-1, 1], (channels, samples))
    <key>.times.csv         ``time[,label]`` lines: annotated onsets                       (``_decode_csv``, :19-25)
    <key>.times.pred.csv    optional: onsets predicted by the video onset model

``iter_shard_samples`` yields the decoded dictionaries, ``sfx_chunks`` composes the reference's pipeline on top of them
(resample to the model rate -- on the device with the library's sinc resampler when a device is given --, tuple, ``_get_slices``)
and ``sfx_batches`` adds ``collate_fn``.  PCM wav (8 / 16 / 24 / 32-bit integer) is decoded with :mod:`wave`; other encodings
raise (the reference relies on torchaudio for them).
"""
from __future__ import annotations

import io
import random
import tarfile
import wave
from glob import glob
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import torch

from .input_pipeline import collate_fn, slice_chunks

Tensor = torch.Tensor


def decode_wav(data: bytes) -> Tuple[Tensor, int]:
    """PCM wav bytes -> ((channels, samples) float32 in [-1, 1], sample_rate), as ``torchaudio.load(normalize=True)`` scales it."""
    try:
        with wave.open(io.BytesIO(data), "rb") as w:
            ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
    except wave.Error as e:
        raise ValueError(f"unsupported wav encoding (only integer PCM is decoded here): {e}") from e
    if width == 1:
        x = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(torch.float32) - 128.0) / 128.0
    elif width == 2:
        x = torch.frombuffer(bytearray(raw), dtype=torch.int16).to(torch.float32) / 32768.0
    elif width == 3:
        b = torch.frombuffer(bytearray(raw), dtype=torch.uint8).reshape(-1, 3).to(torch.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = torch.where(v >= (1 << 23), v - (1 << 24), v)
        x = v.to(torch.float32) / float(1 << 23)
    elif width == 4:
        x = torch.frombuffer(bytearray(raw), dtype=torch.int32).to(torch.float32) / float(1 << 31)
    else:
        raise ValueError(f"unsupported sample width {width}")
    return x.reshape(-1, ch).t().contiguous(), sr


def decode_times_csv(data: bytes) -> Dict[float, Optional[str]]:
    """main/dataset_diffusion.py:19-25: ``time[,label]`` per line -> {time: label or None}; the text after the last newline is dropped."""
    rows = [line.split(",") for line in data.decode("utf-8").split("\n")[:-1]]
    return {float(r[0]): (r[1] if len(r) > 1 else None) for r in rows}


def _split_key(name: str) -> Tuple[str, str]:
    """WebDataset's grouping rule: the key is the path up to the first dot of the base name, the rest is the extension."""
    head, _, base = name.rpartition("/")
    stem, dot, ext = base.partition(".")
    return (head + "/" + stem if head else stem), ext if dot else ""


def expand_shards(path: Union[str, Sequence[str]]) -> List[str]:
    """A path, a glob pattern, a ``{000..012}`` brace range or a list of those -> sorted shard paths."""
    if not isinstance(path, str):
        out: List[str] = []
        for p in path:
            out += expand_shards(p)
        return out
    if "{" in path and ".." in path:
        pre, rest = path.split("{", 1)
        rng, post = rest.split("}", 1)
        lo, hi = rng.split("..")
        return [f"{pre}{str(i).zfill(len(lo))}{post}" for i in range(int(lo), int(hi) + 1)]
    hits = sorted(glob(path))
    return hits if hits else [path]


def iter_shard_samples(path: Union[str, Sequence[str]], shardshuffle: bool = False, rng: Optional[random.Random] = None) -> Iterator[dict]:
    """Decoded samples of the shards in order: ``{"__key__", "resampled.wav": (tensor, sr), "times.csv": {...}, ...}``."""
    shards = expand_shards(path)
    if shardshuffle:
        (rng or random).shuffle(shards)
    for shard in shards:
        with tarfile.open(shard, "r:*") as tf:
            cur: dict = {}
            for m in tf:
                if not m.isfile():
                    continue
                key, ext = _split_key(m.name)
                if cur and key != cur["__key__"]:
                    yield cur
                    cur = {}
                cur.setdefault("__key__", key)
                data = tf.extractfile(m).read()
                if ext.endswith("wav"):
                    cur[ext] = decode_wav(data)
                elif ext.endswith("csv"):
                    cur[ext] = decode_times_csv(data)
                else:
                    cur[ext] = data
            if cur:
                yield cur


def sfx_chunks(path: Union[str, Sequence[str]], sample_rate: int, chunk_size: int, shardshuffle: bool = False, shift_augment: bool = False,
               cut_prefix: bool = True, one_chunk_per_track: bool = True, onset_check_length: Optional[int] = None, device=None,
               rng: Optional[random.Random] = None) -> Iterator[Tuple[Tensor, Tensor, Tensor, str, str]]:
    """``create_sfx_dataset`` (main/dataset_diffusion.py:111-131): shards -> decoded samples -> resampled to ``sample_rate`` ->
    ``(wav_chunk, pred_onset_chunk, cond_chunk, text, filename)``.  With ``device`` (a CUDA device) the sinc resampling of the
    source audio runs in the HIP library (``sf_resampler_*``, the algorithm torchaudio's ``resample`` uses) and the chunks stay there."""
    for sample in iter_shard_samples(path, shardshuffle, rng):
        wav, sr = sample["resampled.wav"]                                   # _to_tuple (:28-33)
        if device is not None:
            wav = wav.to(device)
        if sr != sample_rate:                                               # _fn_resample (:15-16)
            if device is None:
                raise ValueError(f"{sample['__key__']}: stored at {sr} Hz, needs {sample_rate} Hz: pass a CUDA device (the resampler is a HIP kernel)")
            from .resample import resample

            wav = resample(wav[None], orig_freq=sr, new_freq=sample_rate)[0]
        yield from slice_chunks(wav, sample_rate, sample["times.csv"], sample.get("times.pred.csv"), sample["__key__"], chunk_size,
                                onset_check_length, shift_augment, cut_prefix, one_chunk_per_track, rng)


def sfx_batches(path: Union[str, Sequence[str]], batch_size: int, **kwargs) -> Iterator[tuple]:
    """Collated batches ``(x, y, z, texts, filenames)`` as the reference's DataLoader yields them (main/dataset_diffusion.py:134-143;
    main/datamodule_diffusion.py:36-44); the last batch may be smaller."""
    buf: list = []
    for item in sfx_chunks(path, **kwargs):
        buf.append(item)
        if len(buf) == batch_size:
            yield collate_fn(buf)
            buf = []
    if buf:
        yield collate_fn(buf)
