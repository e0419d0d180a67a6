"""VideoOnsetNet throughput for 32 clips processed in sub-batches (does the 256 MB Infinity Cache hold a sub-batch's tensors?):
python tools/onset_chunks.py [dtype]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from syncfusion_amd import VideoOnsetNet
from syncfusion_amd import workmodel
torch.set_grad_enabled(False)
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = VideoOnsetNet(False, dtype=dtype).to(dev).eval()
N = 32
x = torch.randn(N, 3, 30, 112, 112, device=dev)
for sub in (32, 16, 8, 4, 2, 1):
    parts = [x[i:i + sub].contiguous() for i in range(0, N, sub)]
    for p in parts[:2]:
        net(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ys = [net(p) for p in parts]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"sub-batch {sub:2d}: {dt * 1e3:7.2f} ms per 32 clips  {N / dt:7.1f} clips/s  {N * workmodel.ONSET_NET_GFLOP_PER_CLIP / 1e3 / dt:6.1f} TFLOP/s", flush=True)
