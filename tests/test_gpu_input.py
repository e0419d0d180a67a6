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
