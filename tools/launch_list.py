#!/usr/bin/env python3
"""Per-launch list of one U-Net evaluation (HIP events around every launch: label, us, algorithmic MFLOP and MB, depth), slowest first.
    python3 tools/launch_list.py B scale dtype [top]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, scale, dtype = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
sys.argv = sys.argv[:1]
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dtype, dev)
net = model.model.net
nz = torch.randn(B, 1, bench.L0, generator=torch.Generator().manual_seed(1000)).to(dev)
ch, e = bench.synthetic_conditioning(model, B, bench.L0, dev, real=True)
sig = torch.full((B,), 0.5, device=dev)
with torch.no_grad():
    net.engine().profile_forward(nz, sig, ch, e, scale)
    recs = net.engine().profile_forward(nz, sig, ch, e, scale, with_depth=True)
tot = sum(r[1] for r in recs)
print(f"{len(recs)} launches, {tot:.3f} ms of device time (serialised)")
for i, r in sorted(enumerate(recs), key=lambda t: -t[1][1])[:top]:
    print(f"{i:4d} d{r[4]:<2d} {r[0]:34s} {r[1] * 1e3:7.2f} us {r[2] / 1e6:9.1f} MFLOP {r[3] / 1e6:7.2f} MB")
