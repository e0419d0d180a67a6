#!/usr/bin/env python3
"""Which Python call sites issue the aten fill / add / sum launches of one training step (torch profiler, with_stack)?
    python tools/train_op_probe.py [--length 65536]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import syncfusion_amd as sa
from syncfusion_amd.reference_config import model_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--length", type=int, default=65536)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = sa.instantiate(model_config()).to(dev)
opt = model.configure_optimizers()
g = torch.Generator().manual_seed(1)
x = torch.randn(args.batch, 1, args.length, generator=g).to(dev)
y = (torch.rand(args.batch, 1, args.length, generator=g) < 0.0005).float().to(dev)
batch = (x, y, x, None, None)
for it in range(2):
    loss = model.training_step(batch, it); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    loss = model.training_step(batch, 2); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::sum", "aten::copy_", "aten::cat", "aten::mul")
by = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        tot[ev.name] += 1
        site = "?"
        for fr in ev.stack:
            if "syncfusion_amd" in fr or "torch/optim" in fr or "autograd/function" in fr or "autograd/graph" in fr:
                site = fr.strip()
                break
        by[ev.name][site[-110:]] += 1
for name in want:
    print(f"== {name}: {tot[name]}")
    for site, n in by[name].most_common(8):
        print(f"   {n:5d}  {site}")
