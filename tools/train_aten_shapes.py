"""ATen ops of one training step by (op, input shapes) with device time: which torch plumbing is worth folding into the HIP ops.
    python tools/train_aten_shapes.py [length]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import syncfusion_amd as sa
from syncfusion_amd.reference_config import model_config

L = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", None)
    if t is None:
        t = getattr(ev, "self_cuda_time_total", 0)
    if ev.key.startswith("aten::") and t > 0:
        rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"ATen device time in one step: {tot / 1e3:.2f} ms")
for t, n, k, sh in rows[:45]:
    print(f"{t / 1e3:7.3f} ms {n:4d}  {k:28s} {sh}")
