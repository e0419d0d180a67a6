"""How fast can this box dispatch dependent tiny kernels? (eager vs hipGraph replay)"""
import time, torch
dev = torch.device("cuda")
x = torch.zeros(1024, device=dev)
N = 300
def chain():
    for _ in range(N):
        x.add_(1.0)
chain(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): chain()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"eager: {dt / N * 1e6:.2f} us per tiny kernel")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    chain(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"graph: {dt / N * 1e6:.2f} us per tiny kernel")
