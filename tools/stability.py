"""Repeated sample() calls alternating step counts and guidance (graph cache hits, misses and re-captures): every repeat of a
configuration must reproduce its first result bit for bit.  python tools/stability.py"""
import contextlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
torch.set_grad_enabled(False)   # inference tools: with autograd recording the modules switch to the training composition
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(sys.stderr):
    model = bench.build_model("bf16", dev)
B, L0 = 8, bench.L0
noise = torch.randn(B, 1, L0, device=dev)
y = torch.zeros(B, 1, L0, device=dev); y[:, 0, ::2205] = 1.0
_, info = model.onsets_encoder(y, with_info=True)
ch = info["xs"][2:-1]
emb = torch.randn(B, 1, 512, device=dev)
ref = None
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
for i in range(12):
    scale = 1.0 if i % 3 else 7.5
    steps = 150 if i % 2 == 0 else 40
    out = model.model.sample(x_noisy=noise, num_steps=steps, channels=ch, embedding=emb, embedding_scale=scale)
    key = (steps, scale)
    ref = ref or {}
    if key in ref: assert torch.equal(ref[key], out), f"call {i} differs"
    else: ref[key] = out.clone()
torch.cuda.synchronize()
print("12 calls ok in %.2f s; free memory change: %.1f MB" % (time.perf_counter() - t0, (free0 - torch.cuda.mem_get_info()[0]) / 1e6))
