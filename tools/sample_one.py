#!/usr/bin/env python3
"""One sample() call at a chosen shape, for profiling:  python tools/sample_one.py B scale steps [dtype] [L0]
Prints ms/step of the second call (graph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B, scale, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
L0 = int(sys.argv[5]) if len(sys.argv) > 5 else bench.L0
dev = torch.device("cuda", 0)
with torch.no_grad():
    model = bench.build_model(dtype, dev)
    if os.environ.get("SF_NO_GRAPH"):
        model.model.sampler.use_graph = False      # per-kernel counters (rocprofv3 --pmc) need eager launches
    noise = torch.randn(B, 1, L0, device=dev)
    y = torch.zeros(B, 1, L0, device=dev); y[:, 0, ::2205] = 1.0
    _, info = model.onsets_encoder(y, with_info=True)
    ch = info["xs"][2:-1]
    emb = torch.nn.functional.normalize(torch.randn(B, 1, 512, device=dev), dim=-1)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.model.sample(x_noisy=noise, num_steps=steps, channels=ch, embedding=emb, embedding_scale=scale)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B} scale={scale} steps={steps} {dtype} L0={L0}: {dt*1e3/steps:.3f} ms/step = {steps/dt:.1f} steps/s")
