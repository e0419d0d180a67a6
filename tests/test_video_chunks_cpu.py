"""Frame-directory dataset mirror (syncfusion_amd/video_chunks.py) on the CPU: chunk table, natural ordering, image decoding."""
import json
import os

import numpy as np
import torch

from syncfusion_amd import video_chunks as vc


def _make_video(root, name, n_frames, frame_rate, duration, times, size=(24, 32)):
    d = root / name / "frames"
    d.mkdir(parents=True)
    from PIL import Image

    rs = np.random.RandomState(len(name))
    imgs = []
    for i in range(n_frames):
        a = rs.randint(0, 256, size=(size[0], size[1], 3), dtype=np.uint8)
        Image.fromarray(a).save(d / f"frame_{i + 1}.png")
        imgs.append(a)
    (root / name / f"{name}.metadata.json").write_text(json.dumps({"processed": {"video_frame_rate": frame_rate, "video_duration": duration}}))
    (root / name / f"{name}.times.csv").write_text("".join(f"{t},hit\n" for t in times))
    return imgs


def test_natural_order_and_chunk_table(tmp_path):
    assert vc.natural_sorted(["f_10.jpg", "f_2.jpg", "f_1.jpg", "f_11.jpg"]) == ["f_1.jpg", "f_2.jpg", "f_10.jpg", "f_11.jpg"]
    _make_video(tmp_path, "vidA", 14, 6.0, 2.4, [0.1, 0.99, 1.5, 2.2])
    table = vc.chunk_table(str(tmp_path), ["vidA"], chunk_length_in_seconds=1.0)
    assert len(table) == 2                                              # int(2.4 / 1.0) whole chunks; the onset at 2.2 s is dropped
    assert [(c["start_frame"], c["end_frame"]) for c in table] == [(0, 6), (6, 12)]
    assert table[0]["labels"].tolist() == [1, 0, 0, 0, 0, 1]            # int(0.1 * 6) = 0, int(0.99 * 6) = 5
    assert table[1]["labels"].tolist() == [0, 0, 0, 1, 0, 0]            # int(0.5 * 6) = 3


def test_frames_are_read_in_natural_order(tmp_path):
    imgs = _make_video(tmp_path, "vidB", 12, 6.0, 2.0, [0.5])
    import glob

    chunk = vc.chunk_table(str(tmp_path), ["vidB"], chunk_length_in_seconds=1.0)[1]
    paths = vc.natural_sorted(glob.glob(os.path.join(chunk["frames_path"], "*.png")))[chunk["start_frame"]: chunk["end_frame"]]
    assert [os.path.basename(p) for p in paths] == [f"frame_{i}.png" for i in range(7, 13)]
    u8 = vc.read_frames_u8(paths, pin=False)
    assert u8.dtype == torch.uint8 and u8.shape == (6, 24, 32, 3)
    assert torch.equal(u8, torch.from_numpy(np.stack(imgs[6:12])))
