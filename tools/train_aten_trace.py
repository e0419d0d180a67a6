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
import traceback
from torch.utils._python_dispatch import TorchDispatchMode


class Trace(TorchDispatchMode):
    """every ATen call of one step with the innermost syncfusion_amd / tools source line that issued it (torch.profiler's with_stack gives
    empty stacks on this build); calls made by C++ autograd nodes carry no Python frame and are listed by shape"""

    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(k in name for k in ("empty", "view", "reshape", "as_strided", "transpose", "slice", "select", "unsqueeze", "squeeze", "expand", "detach", "alias", "permute", "t.default", "_unsafe_view", "split", "unbind", "narrow", "sym_", "is_", "size", "stride", "numel", "item", "_local_scalar")):
            return out
        where = None
        chain = []
        for fr in reversed(traceback.extract_stack()[:-1]):
            if ("syncfusion_amd" in fr.filename or "tools/" in fr.filename) and "train_aten_trace" not in fr.filename:
                chain.append(f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}")
                if len(chain) == 3:
                    break
        if chain:
            where = " <- ".join(chain)
            if "clone" in name or "copy_" in name:
                t = next((a for a in args if isinstance(a, torch.Tensor)), None)
                if t is not None:
                    where += f"  {tuple(t.shape)} strides {tuple(t.stride())}"
        if where is None:
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            where = f"<autograd engine / torch> {shp}"
        self.agg[(name, where)] += 1
        return out


with Trace() as tr:
    step(2)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, where), n in tr.agg.items():
    tot[name] += n
print("== ATen calls of one training step by op ==")
for name, n in tot.most_common(30):
    print(f"{n:5d}  {name}")
print("== by (op, source line) ==")
for (name, where), n in tr.agg.most_common(90):
    print(f"{n:5d}  {name:24s} {where}")
