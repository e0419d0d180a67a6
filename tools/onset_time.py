"""VideoOnsetNet forward time at N clips (python tools/onset_time.py [N] [dtype]): prints 'X ms = Y TFLOP/s' (293.2 GFLOP per clip)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from syncfusion_amd import VideoOnsetNet
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = VideoOnsetNet(False, dtype=dtype).to(dev).eval()
x = torch.randn(N, 3, 30, 112, 112, device=dev)
for _ in range(2):
    y = net(x)
torch.cuda.synchronize()
t = time.perf_counter()
reps = 8
for _ in range(reps):
    y = net(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"{dt * 1e3:.2f} ms = {293.2e9 * N / dt / 1e12:.1f} TFLOP/s")
