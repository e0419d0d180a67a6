"""One VideoOnsetNet forward loop at N clips for rocprofv3 (python tools/onset_one.py [N] [dtype] [reps])."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from syncfusion_amd import VideoOnsetNet
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = VideoOnsetNet(False, dtype=dtype).to(dev).eval()
x = torch.randn(N, 3, 30, 112, 112, device=dev)
for _ in range(reps):
    y = net(x)
torch.cuda.synchronize()
print(float(y.sum()))
