#!/usr/bin/env python3
"""Is the FIRST 20-step window of a process slower than the later ones?  Fresh process, bench.py's own model and inputs:
setup run(2), warm-up run(5), then N timed 20-step windows back to back; optionally a pre-roll of P steps before the warm-up.
    python3 tools/headline_windows.py [preroll_steps] [windows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sys.argv = sys.argv[:1]
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
with torch.no_grad():
    model = bench.build_model("bf16", dev)
    noise = torch.randn(8, 1, bench.L0, generator=torch.Generator().manual_seed(1000)).to(dev)
    ch, e = bench.synthetic_conditioning(model, 8, bench.L0, dev, real=False)
    run = lambda n: model.model.sample(x_noisy=noise, num_steps=n, channels=ch, embedding=e, embedding_scale=1.0)  # noqa: E731
    run(2)
    if pre:
        run(pre)
    run(5)
    rates = []
    for _ in range(nwin):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(20)
        torch.cuda.synchronize(dev)
        rates.append(20 / (time.perf_counter() - t0))
print(f"preroll {pre:4d}: " + " ".join(f"{r:6.1f}" for r in rates))
