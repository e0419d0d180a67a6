"""Device side of the reference's denoising driver, ``main.generation.generate_dataset``.

Reference loop body (main/generation.py:49-103): ``noise = randn(B,1,length)`` ->
``onsets_encoder(y, with_info=True)`` -> CLAP embedding -> ``model.model.sample(...)`` -> per clip:
zero everything before the first onset (``cut_prefix``), crop to ``cut_length``, resample, save.
``generate_batch`` is that body up to (not including) the CPU resample/save, with the same keyword
names; the DataLoader / WebDataset / torchaudio.save plumbing around it is out of scope (SURVEY.md
section 2 rows 5 and 8f-2).  ``generate_dataset`` takes what the reference passes it -- an UN-batched dataset of
``(x, y, z, text, filename)`` chunks (``exp/evaluate_gh_gen.yaml:21,31-40``: ``create_sfx_dataset(...)`` + ``batch_size``) -- and
batches it itself with ``batch_size`` / ``num_workers`` + ``collate_fn`` as main/generation.py:37-38 does; an iterable of
already-collated batches is passed through.  Output files are 32-bit IEEE-float wav (RIFF format tag 3) written with the
standard library: what ``torchaudio.save`` of a float32 tensor (:104-122) produces -- samples are stored bit-exactly, nothing is
clamped or quantised.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Iterable, List, Optional, Sequence, Union

import torch

from . import _lib
from .input_pipeline import collate_fn
from .onset_glue import cut_prefix_crop
from .resample import resample

Tensor = torch.Tensor


@torch.no_grad()
def generate_batch(model, y: Tensor, z: Optional[Tensor] = None, text: Optional[Sequence[str]] = None, *,
                   num_steps: int = 150, length: int = 2 ** 18, embedding_scale: float = 7.5, cut_prefix: bool = False,
                   cond_text: bool = False, cut_length: Optional[int] = None, noise: Optional[Tensor] = None,
                   generator: Optional[torch.Generator] = None) -> Tensor:
    """One batch of main/generation.py:69-89,100 on the device of ``model``.  Returns ``(B, 1, cut_length)``."""
    device = model.device
    _lib.require_gpu_tensor(torch.empty(0, device=device), "generate_batch")
    B = y.shape[0]
    if noise is None:
        noise = torch.randn((B, 1, length), device=device, generator=generator)   # the ONLY RNG on the path (:69)
    y = y.to(device)
    _, y_latent = model.onsets_encoder(y, with_info=True)                        # :71
    if cond_text:
        z_latent = model.clap_encode_text(list(text))                             # :73
    else:
        z_latent = model.clap_encode_audio(z.to(device))                          # :75
    gen = model.model.sample(x_noisy=noise.to(device), num_steps=num_steps, channels=y_latent["xs"][2:-1],
                             embedding=z_latent.to(device), embedding_scale=embedding_scale)   # :77-83
    if cut_prefix:
        # one device pass for the batch (:86-89, :100); IndexError on a track without onsets, as the reference's [0] does
        return cut_prefix_crop(gen, y[:, :1], cut_length or length)
    return gen[:, :, : (cut_length or length)]                                     # :91,100


def save_wav(path: Union[str, Path], audio: Tensor, sample_rate: int) -> None:
    """(channels, n) float tensor -> 32-bit IEEE-float wav, the file ``torchaudio.save(path, float32 tensor, rate)`` writes
    (main/generation.py:104-122): RIFF/WAVE, ``fmt `` with format tag 3 (WAVE_FORMAT_IEEE_FLOAT), the ``fact`` chunk non-PCM
    formats carry, interleaved little-endian float32 frames.  Samples keep their bits: no clamp, no rounding."""
    a = audio.detach().to(torch.float32).cpu()
    if a.dim() != 2:
        raise ValueError(f"save_wav expects (channels, samples), got shape {tuple(a.shape)}")
    ch, n = int(a.shape[0]), int(a.shape[1])
    data = a.t().contiguous().numpy().astype("<f4", copy=False).tobytes()
    rate = int(sample_rate)
    fmt = struct.pack("<HHIIHH", 3, ch, rate, rate * ch * 4, ch * 4, 32)
    body = (b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<II", 4, n) +
            b"data" + struct.pack("<I", len(data)) + data)
    with open(str(path), "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def load_wav(path: Union[str, Path]):
    """Reader for the files ``save_wav`` writes (and 16-bit PCM wav): -> ((channels, n) float32 tensor, sample_rate).
    The standard library's ``wave`` module rejects format tag 3."""
    raw = Path(path).read_bytes()
    if raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", raw[pos + 8:pos + 24])
        elif cid == b"data":
            data = raw[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt / data chunk")
    tag, ch, rate, _, _, bits = fmt
    import numpy as np

    if tag == 3 and bits == 32:
        a = torch.from_numpy(np.frombuffer(data, dtype="<f4").astype(np.float32))
    elif tag == 1 and bits == 16:
        a = torch.from_numpy(np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0)
    else:
        raise ValueError(f"{path}: unsupported wav encoding (format tag {tag}, {bits} bits)")
    return a.reshape(-1, ch).t().contiguous(), int(rate)


def _is_collated(elem) -> bool:
    """A collated batch carries (B, C, T) waveforms and a sequence of filenames; a dataset element (C, T) and one filename."""
    x = elem[0]
    return isinstance(x, torch.Tensor) and x.dim() == 3 and not isinstance(elem[4], (str, bytes))


def iter_batches(dataset: Iterable, batch_size: int, num_workers: int = 0) -> Iterable:
    """What ``DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, collate_fn=collate_fn)`` yields
    (main/generation.py:37-38).  torch ``Dataset`` / ``IterableDataset`` objects go through that very DataLoader; any other
    iterable of un-collated ``(x, y, z, text, filename)`` chunks is grouped here in order (last batch short, as the
    DataLoader's ``drop_last=False``); an iterable whose elements already are collated batches is passed through."""
    if batch_size < 1:
        raise ValueError(f"batch_size must be >= 1, got {batch_size}")
    if isinstance(dataset, (torch.utils.data.Dataset, torch.utils.data.IterableDataset)):
        yield from torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, collate_fn=collate_fn)
        return
    it = iter(dataset)
    try:
        first = next(it)
    except StopIteration:
        return
    if _is_collated(first):
        yield first
        yield from it
        return
    buf = [first]
    for elem in it:
        if len(buf) == batch_size:
            yield collate_fn(buf)
            buf = []
        buf.append(elem)
    if buf:
        yield collate_fn(buf)


@torch.no_grad()
def generate_dataset(experiment_path: Union[str, Path], model, dataset: Iterable, device: str = "cuda",
                     model_path: Optional[str] = None, batch_size: int = 16, num_workers: int = 4, sample_rate: int = 48000,
                     num_steps: int = 150, length: int = 2 ** 18, embedding_scale: float = 7.5, cut_prefix: bool = False,
                     cond_text: bool = False, one_chunk_per_track: bool = False, cut_length: Optional[int] = None,
                     downsample_rate: Optional[int] = None, save_cond: bool = False) -> List[Path]:
    """Same signature as main/generation.py:12-30.  ``dataset``: the reference's un-batched chunk dataset (batched here with
    ``batch_size`` / ``num_workers`` + ``collate_fn``, :37-38) or an iterable of already-collated batches (``iter_batches``)."""
    experiment_path = Path(experiment_path)
    experiment_path.mkdir(exist_ok=True, parents=True)
    if model_path:                                                                  # :40-44
        checkpoint = torch.load(model_path, map_location=device)
        model.load_state_dict(checkpoint["state_dict"])     # upstream or local key layout (syncfusion_amd/keymap.py)
    model.to(device)
    written: List[Path] = []
    chunk_id = 0
    for batch in iter_batches(dataset, batch_size, num_workers):
        x, y, z, text, filenames = batch
        B = x.shape[0]
        if not one_chunk_per_track:
            last = experiment_path / f"{chunk_id + B - 1}.wav"
            if last.exists():                                                       # crude resume, :52-59
                chunk_id += B
                continue
        gen = generate_batch(model, y, z, text, num_steps=num_steps, length=length, embedding_scale=embedding_scale,
                             cut_prefix=cut_prefix, cond_text=cond_text, cut_length=cut_length)
        out_sr = sample_rate
        cond = z.to(gen.device).to(torch.float32) if (save_cond and not cond_text) else None   # :93-96: the conditioning audio itself
        if downsample_rate:                                                       # :90-98, on the device instead of the CPU
            gen = resample(gen, orig_freq=sample_rate, new_freq=downsample_rate)
            if cond is not None:
                cond = resample(cond, orig_freq=sample_rate, new_freq=downsample_rate)
            out_sr = downsample_rate
        for i in range(B):
            stem = f"{chunk_id}" if not one_chunk_per_track else str(filenames[i]).split('/')[-1]
            # file set of :104-122: `{stem}.wav`; with save_cond `{stem}_{text}.wav` (text conditioning, replaces the plain name)
            # or `{stem}.wav` + `{stem}_cond.wav` (audio conditioning)
            name = f"{stem}_{text[i]}.wav" if (save_cond and cond_text) else f"{stem}.wav"
            save_wav(experiment_path / name, gen[i], out_sr)
            written.append(experiment_path / name)
            if cond is not None:
                save_wav(experiment_path / f"{stem}_cond.wav", cond[i], out_sr)
                written.append(experiment_path / f"{stem}_cond.wav")
            chunk_id += 1
    return written
