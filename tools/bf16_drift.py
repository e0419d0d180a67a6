"""bf16 vs fp32 engine over a whole sampling loop (same weights, noise and conditioning): how far the bf16 path drifts.
python tools/bf16_drift.py"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.set_grad_enabled(False)   # inference tools: with autograd recording the modules switch to the training composition
import bench

dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(sys.stderr):
    m32 = bench.build_model("fp32", dev)
    m16 = bench.build_model("bf16", dev)
m16.load_state_dict(m32.state_dict())
B, L0 = 2, bench.L0
g = torch.Generator().manual_seed(7)
noise = torch.randn(B, 1, L0, generator=g).to(dev)
y = torch.zeros(B, 1, L0, device=dev)
y[:, 0, ::4410] = 1.0
emb = torch.randn(B, 1, 512, generator=g).to(dev)
_, info = m32.onsets_encoder(y, with_info=True)
ch = info["xs"][2:-1]
for steps, scale in ((1, 1.0), (10, 1.0), (50, 1.0), (50, 7.5), (150, 7.5)):
    a = m32.model.sample(x_noisy=noise, num_steps=steps, channels=ch, embedding=emb, embedding_scale=scale)
    b = m16.model.sample(x_noisy=noise, num_steps=steps, channels=ch, embedding=emb, embedding_scale=scale)
    rel = float((a - b).norm() / a.norm())
    print(f"steps={steps:4d} scale={scale}: rel-L2(bf16 vs fp32) = {rel:.3e}   rms(out) = {float(a.pow(2).mean().sqrt()):.3f}", flush=True)
