"""Input side on the device: frame transform and impulse track against their oracles."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,T,H,W", [(2, 30, 240, 320), (1, 5, 112, 112), (3, 4, 100, 130), (1, 3, 64, 48)])
def test_frames_to_clip_matches_reference_transform(cuda, N, T, H, W):
    """320x240 is what script/gh_preprocess_videos.py:143-144 extracts; 112x112 = identity resize; odd sizes; upscaling."""
    from oracle import frames_ref
    from syncfusion_amd.input_pipeline import frames_to_clip

    u8 = torch.randint(0, 256, (N, T, H, W, 3), generator=torch.Generator().manual_seed(H), dtype=torch.uint8)
    ref = frames_ref.frames_transform(u8)
    got = frames_to_clip(u8.to(cuda))
    assert got.shape == (N, 3, T, 112, 112)
    assert float((got.cpu() - ref).abs().max()) < 2e-5
    with pytest.raises(ValueError):
        frames_to_clip(u8.to(cuda).float())


def test_times_to_track_matches_python_int_truncation(cuda):
    from syncfusion_amd.input_pipeline import times_to_track

    sr, L = 48000, 96000
    times = [[0.0, 0.1234, 1.99999], [0.5], [], [1.0000001, 2.5]]          # 2.5 s falls outside the 2 s chunk and is dropped
    tr = times_to_track(times, sr, L, cuda)
    assert tr.shape == (4, 1, L)
    for b, ts in enumerate(times):
        want = sorted({int(t * sr) for t in ts if int(t * sr) < L})
        assert torch.nonzero(tr[b, 0]).flatten().tolist() == want
    assert float(tr.sum()) == 5.0


def test_pinned_prefetcher_feeds_generate_batch_inputs(cuda):
    from syncfusion_amd.input_pipeline import PinnedPrefetcher

    g = torch.Generator().manual_seed(0)
    batches = [(torch.randn(2, 1, 64, generator=g), torch.zeros(2, 1, 64), torch.randn(2, 1, 16 + i, generator=g), ["a", "b"], [f"f{i}", f"g{i}"]) for i in range(5)]
    got = list(PinnedPrefetcher(batches, cuda))
    assert len(got) == 5
    for (x, y, z, text, fn), (gx, gy, gz, gtext, gfn) in zip(batches, got):
        torch.cuda.synchronize()
        assert gx.is_cuda and torch.equal(gx.cpu(), x) and torch.equal(gz.cpu(), z) and gtext == text and gfn == fn


def test_shards_to_batches_with_device_resampling(cuda, tmp_path):
    """Shard (8 kHz PCM) -> create_sfx_dataset's pipeline at 16 kHz: the source audio is resampled by the HIP sinc resampler
    (torchaudio's algorithm, main/dataset_diffusion.py:15-16) and the chunks stay on the device."""
    import io
    import random
    import tarfile
    import wave

    import numpy as np

    from syncfusion_amd import resample, shards
    from syncfusion_amd.input_pipeline import slice_chunks

    sr, n = 8000, 6000
    x = (np.random.RandomState(1).randn(1, n) * 4000).astype(np.int16)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(x.T.astype("<i2").tobytes())
    path = tmp_path / "s-000.tar"
    with tarfile.open(path, "w") as tf:
        for name, data in (("k.resampled.wav", buf.getvalue()), ("k.times.csv", b"0.1,tap\n0.4,tap\n")):
            ti = tarfile.TarInfo(name)
            ti.size = len(data)
            tf.addfile(ti, io.BytesIO(data))
    got = list(shards.sfx_chunks(str(path), sample_rate=16000, chunk_size=4096, one_chunk_per_track=False, device=cuda, rng=random.Random(0)))
    wav16 = resample(torch.from_numpy(x.astype(np.float32) / 32768.0).to(cuda)[None], orig_freq=sr, new_freq=16000)[0]
    want = list(slice_chunks(wav16, 16000, {0.1: "tap", 0.4: "tap"}, None, "k", 4096, one_chunk_per_track=False, rng=random.Random(0)))
    assert len(got) == len(want) == 2 and got[0][0].is_cuda and got[0][0].shape == (1, 4096)
    for a, b in zip(got, want):
        assert all(torch.equal(u, v) for u, v in zip(a[:3], b[:3])) and a[3:] == b[3:]


def test_frame_directory_chunks_to_onset_net_input(cuda, tmp_path):
    """main/dataset_onset.py:121-165: frames of a chunk (naturally sorted files, PIL RGB decode) -> (3, T, 112, 112); the transform
    chain against its restatement (oracle/frames_ref.py, pinned to the ATen antialias kernel)."""
    import json

    import numpy as np
    from PIL import Image

    from oracle import frames_ref
    from syncfusion_amd import video_chunks as vc

    d = tmp_path / "v1" / "frames"
    d.mkdir(parents=True)
    rs = np.random.RandomState(5)
    imgs = []
    for i in range(9):
        a = rs.randint(0, 256, size=(60, 80, 3), dtype=np.uint8)
        Image.fromarray(a).save(d / f"{i + 1}.jpg", quality=95)
        imgs.append(np.asarray(Image.open(d / f"{i + 1}.jpg").convert("RGB")))     # what a JPEG decoder returns
    (tmp_path / "v1" / "v1.metadata.json").write_text(json.dumps({"processed": {"video_frame_rate": 4.0, "video_duration": 2.1}}))
    (tmp_path / "v1" / "v1.times.csv").write_text("0.3,hit\n1.6,hit\n")
    table = vc.chunk_table(str(tmp_path), ["v1"], chunk_length_in_seconds=1.0)
    assert len(table) == 2 and table[1]["labels"].tolist() == [0, 0, 1, 0]
    clips, labels, part = next(vc.iter_clips(table, batch_size=2, device=cuda))
    assert clips.shape == (2, 3, 4, 112, 112) and labels.shape == (2, 4) and len(part) == 2
    want = frames_ref.frames_transform(torch.from_numpy(np.stack(imgs[4:8]))[None])[0]   # chunk 1 = frames 5..8
    assert rel_l2(clips[1].cpu(), want) < 2e-5


def test_shards_feed_generate_dataset(cuda, tmp_path):
    """The reference's evaluation flow end to end on the small model: tar shard -> create_sfx_dataset's batches ->
    generate_dataset (main/generation.py:12-122) -> one wav per chunk, named by the running chunk id."""
    import functools
    import io
    import random
    import tarfile
    import wave

    import numpy as np

    import syncfusion_amd as sa
    from helpers import SMALL_ENCODER, SMALL_UNET
    from syncfusion_amd import shards

    sr, L0 = 8000, 16 * 64
    path = tmp_path / "eval-000.tar"
    rs = np.random.RandomState(2)
    with tarfile.open(path, "w") as tf:
        for k in range(3):
            x = (rs.randn(1, 3 * L0) * 5000).astype(np.int16)
            buf = io.BytesIO()
            with wave.open(buf, "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(sr)
                w.writeframes(x.T.astype("<i2").tobytes())
            for name, data in ((f"clip{k}.resampled.wav", buf.getvalue()), (f"clip{k}.times.csv", b"0.01,hit\n0.2,hit\n")):
                ti = tarfile.TarInfo(name)
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))
    kw = dict(SMALL_UNET)
    model = sa.Model(1e-4, 0.95, 0.999, 1e-6, 1e-3,
                     sa.DiffusionModel(net_t=functools.partial(sa.UNetV0, seed=0), diffusion_t=sa.VDiffusion, sampler_t=sa.VSampler, use_embedding_cfg=True, **kw),
                     sa.Encoder1d(seed=1, **SMALL_ENCODER), sa.RandomEmbedder(kw["embedding_features"]), None).to(cuda)
    batches = list(shards.sfx_batches(str(path), batch_size=2, sample_rate=sr, chunk_size=L0, one_chunk_per_track=True, rng=random.Random(0)))
    assert [b[0].shape[0] for b in batches] == [2, 1]
    files = sa.generate_dataset(tmp_path / "gen", model, batches, device="cuda", sample_rate=sr, num_steps=3, length=L0, embedding_scale=2.0,
                                cut_prefix=True, cut_length=L0 // 2)
    assert [f.name for f in files] == ["0.wav", "1.wav", "2.wav"]
    from syncfusion_amd.generation import load_wav

    got, rate = load_wav(files[2])
    assert rate == sr and got.shape == (1, L0 // 2)
    # The reference hands generate_dataset the UN-batched chunk dataset plus batch_size (exp/evaluate_gh_gen.yaml:21,31-40;
    # main/generation.py:37-38 wraps it in DataLoader(batch_size, num_workers, collate_fn)): same files, same samples.
    chunks = shards.sfx_chunks(str(path), sample_rate=sr, chunk_size=L0, one_chunk_per_track=True, rng=random.Random(0))
    torch.manual_seed(11)
    files_a = sa.generate_dataset(tmp_path / "gen_a", model, batches, device="cuda", batch_size=2, sample_rate=sr, num_steps=3, length=L0,
                                  embedding_scale=2.0, cut_prefix=True, cut_length=L0 // 2)
    torch.manual_seed(11)
    files_b = sa.generate_dataset(tmp_path / "gen_b", model, chunks, device="cuda", batch_size=2, num_workers=0, sample_rate=sr, num_steps=3,
                                  length=L0, embedding_scale=2.0, cut_prefix=True, cut_length=L0 // 2)
    assert [f.name for f in files_b] == ["0.wav", "1.wav", "2.wav"]
    for fa, fb in zip(files_a, files_b):
        assert fa.read_bytes() == fb.read_bytes()
