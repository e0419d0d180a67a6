"""aten-level operator table of one training step (which torch ops surround the HIP Functions):  python tools/train_step_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import syncfusion_amd as sa
from syncfusion_amd.reference_config import model_config

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = sa.instantiate(model_config()).to(dev)
opt = model.configure_optimizers()
B, L = 4, 262144
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, L, generator=g).to(dev)
y = (torch.rand(B, 1, L, generator=g) < 0.0005).float().to(dev)
batch = (x, y, x, None, None)
for _ in range(2):
    loss = model.training_step(batch, 0); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    loss = model.training_step(batch, 0); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60))
