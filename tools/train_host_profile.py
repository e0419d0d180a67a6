#!/usr/bin/env python3
"""Host-side profile of the eager training step (cProfile over 3 steps after warm-up): where the Python time of ~3600 launches goes.
    python3 tools/train_host_profile.py [top]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import syncfusion_amd as sa  # noqa: E402
from syncfusion_amd.reference_config import model_config  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 35
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = sa.instantiate(model_config()).to(dev)
opt = model.configure_optimizers()
L = 262144
g = torch.Generator().manual_seed(5)
x = torch.randn(4, 1, L, generator=g).to(dev)
y = (torch.rand(4, 1, L, generator=g) < 0.0005).float().to(dev)


def step(i):
    loss = model.training_step((x, y, x, None, None), i)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    step(3 + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
