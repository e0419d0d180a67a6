"""Wall time of consecutive sample() calls (first call captures the step graph, later ones replay the cached one)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)   # inference tools: with autograd recording the modules switch to the training composition
import bench

dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev)
B, L0 = 8, bench.L0
noise = torch.randn(B, 1, L0, device=dev)
y = torch.zeros(B, 1, L0, device=dev); y[:, 0, ::2205] = 1.0
_, info = model.onsets_encoder(y, with_info=True)
ch = info["xs"][2:-1]
emb = torch.zeros(B, 1, 512, device=dev)
for steps in (5, 50, 50, 50, 150, 150, 1, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.model.sample(x_noisy=noise, num_steps=steps, channels=ch, embedding=emb, embedding_scale=1.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"steps={steps:4d}  {dt*1e3:8.2f} ms  {dt*1e3/steps:6.3f} ms/step", flush=True)
