"""Which Python lines launch the ATen element-wise / reduce / fill kernels of a training step?  torch.profiler over one step of the
reference's training configuration (reduced length), aggregated by (op, innermost syncfusion_amd / tools source line).
    python tools/train_aten_trace.py [length]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import syncfusion_amd as sa
from syncfusion_amd.reference_config import model_config

L = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step(2)
torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::sum", "aten::add", "aten::add_", "aten::copy_", "aten::mul", "aten::cat", "aten::zeros_like", "aten::clone")
agg = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        where = "?"
        for fr in (ev.stack or []):
            if "syncfusion_amd" in fr or "tools/" in fr or "torch/optim" in fr or "autograd/" in fr:
                where = fr.strip()[-110:]
                break
        agg[(ev.name, where)] += 1
for (name, where), n in agg.most_common(40):
    print(f"{n:5d}  {name:18s} {where}")
