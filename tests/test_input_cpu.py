"""CPU suite: host logic of the input side and the frame-transform oracle."""
import random

import numpy as np
import torch
import torch.nn.functional as F


def test_aa_resize_restatement_matches_aten():
    """oracle.frames_ref.aa_weights (the filter the HIP kernel implements) against torch's antialiased bilinear interpolate."""
    from oracle import frames_ref

    g = torch.Generator().manual_seed(0)
    for H, W in ((240, 320), (112, 112), (100, 130), (64, 48)):
        x = torch.rand(2, 3, H, W, generator=g)
        ref = F.interpolate(x, size=(112, 112), mode="bilinear", antialias=True, align_corners=False).numpy()
        assert np.abs(ref - frames_ref.resize_aa_numpy(x.numpy(), (112, 112))).max() < 2e-5
    u8 = torch.randint(0, 256, (2, 3, 30, 40, 3), generator=g, dtype=torch.uint8)
    out = frames_ref.frames_transform(u8, size=(16, 16))
    assert out.shape == (2, 3, 3, 16, 16) and abs(float(out.mean())) < 1.0


def test_slice_chunks_follows_the_reference_contract():
    from syncfusion_amd.input_pipeline import collate_fn, slice_chunks

    sr, chunk = 100, 200
    wav = torch.arange(1, 1001, dtype=torch.float32).reshape(1, 1000)
    onsets = {0.5: "hit wood", 2.55: "None scratch", 4.1: "hit metal", 9.2: "hit wood"}   # chunk 1 (2.0-4.0 s) has one onset at 2.55
    out = list(slice_chunks(wav, sr, onsets, None, "vid", chunk, rng=random.Random(0), one_chunk_per_track=False))
    # chunks 0 (0.5), 1 (2.55), 2 (4.1), 4 (9.2) hold onsets; chunk 3 (6-8 s) is skipped
    assert len(out) == 4
    w0, p0, c0, text, fn = out[0]
    assert fn == "vid" and text in ("hit wood", "hit metal")             # texts containing 'None' are never chosen
    assert w0.shape == (1, chunk) and p0.shape == (1, chunk)
    assert float(w0[0, :50].abs().max()) == 0.0 and float(w0[0, 50]) == 51.0   # cut_prefix zeroes everything before the first onset
    assert int(torch.nonzero(p0[0])[0]) == 50 and float(p0.sum()) == 1.0
    assert torch.equal(c0, w0[:, 50:])                                   # single onset: the conditioning chunk runs to the chunk's end
    assert float(wav[0, 10]) == 11.0, "the source waveform must not be modified"
    w1, p1, c1, _, _ = out[1]
    k1 = int(2.55 * sr) - 200                                            # Python's int(): 2.55 * 100 = 254.99999999999997 -> 254
    assert k1 == 54 and int(torch.nonzero(p1[0])[0]) == k1 and torch.equal(c1, w1[:, k1:])
    one = list(slice_chunks(wav, sr, onsets, None, "vid", chunk, rng=random.Random(0), one_chunk_per_track=True))
    assert len(one) == 1
    # predicted onsets (times.pred.csv) replace the conditioning track but not the cut / skip logic
    pred = {0.7: None}
    wp, pp, _, _, _ = next(slice_chunks(wav, sr, onsets, pred, "vid", chunk, rng=random.Random(0)))
    assert int(torch.nonzero(pp[0])[0]) == 70 and float(wp[0, 49]) == 0.0 and float(wp[0, 50]) == 51.0
    batch = collate_fn(out[:3])
    assert batch[0].shape == (3, 1, chunk) and batch[1].shape == (3, 1, chunk) and batch[2].shape[0] == 3
    assert batch[2].shape[2] == max(o[2].shape[1] for o in out[:3]) and len(batch[3]) == 3


def test_frame_labels():
    from syncfusion_amd.input_pipeline import frame_labels

    lab = frame_labels([0.1, 1.99, 2.0, 3.5], chunk_start_time=2.0, chunk_length_in_seconds=2.0, frame_rate=15.0)
    assert lab.shape == (30,) and lab.nonzero().flatten().tolist() == [0, 22]     # 2.0 -> frame 0, 3.5 -> int(1.5 * 15) = 22
