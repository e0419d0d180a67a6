"""Where the small ATen launches of one training step come from: (op, python frame) counts for fill_ / zero_ / add / copy_ / cat
(torch.profiler with_stack).   python tools/train_fill_sources.py"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import syncfusion_amd as sa
from syncfusion_amd.reference_config import model_config

L = 262144
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = sa.instantiate(model_config()).to(dev)
opt = model.configure_optimizers()
g = torch.Generator().manual_seed(5)
x = torch.randn(4, 1, L, generator=g).to(dev)
y = (torch.rand(4, 1, L, generator=g) < 0.0005).float().to(dev)
def step(i):
    loss = model.training_step((x, y, x, None, None), i)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for i in range(2):
    step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::add_", "aten::add", "aten::copy_", "aten::cat", "aten::sum", "aten::sub", "aten::mul", "aten::contiguous", "aten::clone")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 0:
        frame = next((f for f in (ev.stack or []) if "syncfusion_amd" in f or "bench" in f), (ev.stack or ["?"])[0] if ev.stack else "?")
        cnt[(ev.name, frame.strip()[-110:])] += 1
for (name, frame), n in cnt.most_common(45):
    print(f"{n:5d} {name:16s} {frame}")
