"""Shard reader (syncfusion_amd/shards.py) on the CPU: a tar shard in the reference's WebDataset layout, written with the standard
library, read back into the tuples `create_sfx_dataset` yields (main/dataset_diffusion.py:111-131)."""
import io
import random
import tarfile
import wave

import numpy as np
import pytest
import torch

from syncfusion_amd import shards
from syncfusion_amd.input_pipeline import slice_chunks


def _wav_bytes(x: np.ndarray, sr: int, width: int = 2) -> bytes:
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(x.shape[0])
        w.setsampwidth(width)
        w.setframerate(sr)
        if width == 2:
            w.writeframes(x.T.astype("<i2").tobytes())
        else:
            v = x.T.astype("<i4")
            w.writeframes(b"".join(int(s).to_bytes(4, "little", signed=True)[:3] for s in v.reshape(-1)))
    return buf.getvalue()


def _add(tf, name, data: bytes):
    ti = tarfile.TarInfo(name)
    ti.size = len(data)
    tf.addfile(ti, io.BytesIO(data))


@pytest.fixture()
def shard(tmp_path):
    sr, n = 8000, 4000
    rs = np.random.RandomState(0)
    path = tmp_path / "sfx-000.tar"
    pcm = {}
    with tarfile.open(path, "w") as tf:
        for i, key in enumerate(["set/a_001", "set/b_002", "c_003"]):
            x = (rs.randn(1, n) * 3000).astype(np.int16)
            pcm[key] = x
            _add(tf, f"{key}.resampled.wav", _wav_bytes(x, sr))
            _add(tf, f"{key}.times.csv", f"0.05,hit\n0.21,None\n{0.3 + 0.01 * i:.2f},scrape\n".encode())
            if i == 1:
                _add(tf, f"{key}.times.pred.csv", b"0.06\n0.2\n")
    return path, sr, n, pcm


def test_samples_are_grouped_and_decoded(shard):
    path, sr, n, pcm = shard
    got = list(shards.iter_shard_samples(str(path)))
    assert [s["__key__"] for s in got] == ["set/a_001", "set/b_002", "c_003"]
    for s in got:
        wav, rate = s["resampled.wav"]
        assert rate == sr and wav.shape == (1, n) and wav.dtype == torch.float32
        assert torch.equal(wav, torch.from_numpy(pcm[s["__key__"]].astype(np.float32) / 32768.0))   # torchaudio's normalisation
        assert list(s["times.csv"].values())[:2] == ["hit", "None"]
    assert "times.pred.csv" in got[1] and got[1]["times.pred.csv"] == {0.06: None, 0.2: None}
    assert "times.pred.csv" not in got[0]


def test_chunks_match_the_slicing_of_the_decoded_samples(shard):
    path, sr, n, _ = shard
    kw = dict(chunk_size=1500, cut_prefix=True, one_chunk_per_track=False)
    got = list(shards.sfx_chunks(str(path), sample_rate=sr, rng=random.Random(3), **kw))
    rng = random.Random(3)
    want = []
    for s in shards.iter_shard_samples(str(path)):
        want += list(slice_chunks(s["resampled.wav"][0], sr, s["times.csv"], s.get("times.pred.csv"), s["__key__"], rng=rng, **kw))
    assert len(got) == len(want) > 0
    for a, b in zip(got, want):
        assert all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3:] == b[3:]
    x, y, z, texts, names = next(shards.sfx_batches(str(path), batch_size=2, sample_rate=sr, rng=random.Random(3), **kw))
    assert x.shape == (2, 1, 1500) and y.shape == (2, 1, 1500) and z.shape[:2] == (2, 1) and len(texts) == len(names) == 2
    with pytest.raises(ValueError, match="needs 16000 Hz"):
        next(shards.sfx_chunks(str(path), sample_rate=16000, chunk_size=1500))


def test_24_bit_pcm_and_shard_patterns(tmp_path):
    x = np.array([[0, 1 << 22, -(1 << 22), (1 << 23) - 1, -(1 << 23)]], dtype=np.int64)
    wav, sr = shards.decode_wav(_wav_bytes(x, 22050, width=3))
    assert sr == 22050 and torch.allclose(wav, torch.tensor([[0.0, 0.5, -0.5, 1 - 2.0 ** -23, -1.0]]))
    assert shards.expand_shards("d/s-{008..011}.tar") == ["d/s-008.tar", "d/s-009.tar", "d/s-010.tar", "d/s-011.tar"]
    assert shards._split_key("a/b/c.times.pred.csv") == ("a/b/c", "times.pred.csv")
