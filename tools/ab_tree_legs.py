#!/usr/bin/env python3
"""The batch-32 legs of bench.py's `extra`, for ONE source tree, each timed several times in one process:
    python tools/ab_tree_legs.py <tree root> [reps] [legs: cfg2,cfg3,ref,e2e]
prints one line per leg: min / median / max steps/s (e2e: seconds per 32 clips).  Used by tools/ab_trees.sh to alternate HEAD with
another checkout (e.g. `git worktree add build/r3tree cafa685` + its own `make`) on the same box.  Only names that exist in both
round-3 and later trees are used (bench.build_model / synthetic_conditioning, model.model.sample, generate_batch)."""
import os
import statistics
import sys
import time

root = os.path.abspath(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
legs = (sys.argv[3] if len(sys.argv) > 3 else "cfg1,cfg2,cfg3,ref,e2e").split(",")
sys.path.insert(0, root)
import torch  # noqa: E402

import bench  # noqa: E402  (the tree's own bench.py and package)

dev = torch.device("cuda", 0)
L0 = bench.L0


def timed(fn, n):
    vals = []
    for _ in range(n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        vals.append(time.perf_counter() - t0)
    return vals


def report(name, rates, unit):
    print(f"{name:5s} min {min(rates):8.3f}  median {statistics.median(rates):8.3f}  max {max(rates):8.3f} {unit}", flush=True)


with torch.no_grad():
    model = bench.build_model("bf16", dev)
    net = model.model.net

    def sample_leg(name, B, L, scale, steps):
        nz = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1000)).to(dev)
        ch, e = bench.synthetic_conditioning(model, B, L, dev, real=True)
        run = lambda n: model.model.sample(x_noisy=nz, num_steps=n, channels=ch, embedding=e, embedding_scale=scale)  # noqa: E731
        run(2)
        run(2)
        ts = timed(lambda: run(steps), reps)
        report(name, [steps / t for t in ts], "steps/s")

    if "cfg1" in legs:
        sample_leg("cfg1", 8, L0, 1.0, 50)
    if "cfg2" in legs:
        sample_leg("cfg2", 32, L0, 2.0, 50)
    if "cfg3" in legs:
        sample_leg("cfg3", 32, L0, 1.0, 50)
    if "ref" in legs:
        sample_leg("ref", 10, 262144, 2.0, 20)
    if "e2e" in legs:
        from syncfusion_amd.generation import generate_batch
        from syncfusion_amd.onset_glue import onsets_to_track
        from syncfusion_amd.onset_net import VideoOnsetNet

        B, steps, scale = 32, 100, 2.0
        torch.manual_seed(7)
        onset = VideoOnsetNet(False, dtype="fp16").to(dev).eval()
        frames = torch.randn(B, 3, 30, 112, 112, generator=torch.Generator().manual_seed(4000)).to(dev)
        z = (torch.randn(B, 1, L0, generator=torch.Generator().manual_seed(1)) * 0.1).to(dev)
        net.compute_dtype = "fp16"

        def once(n_steps):
            logits = onset(frames)
            logits[:, 3] = 1.0
            track = onsets_to_track(logits, L0, frame_rate=15.0, sample_rate=22528.0)
            return generate_batch(model, track, z, num_steps=n_steps, length=L0, embedding_scale=scale, cut_prefix=True, cut_length=44100)

        once(2)
        ts = timed(lambda: once(steps), reps)
        report("e2e", ts, "s per 32 clips")
