"""VideoOnsetNet bf16, N = 32, a few forwards (for rocprofv3 --kernel-trace --stats): python tools/onset_one.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from syncfusion_amd import VideoOnsetNet
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = VideoOnsetNet(False, dtype='bf16').to(dev).eval()
x = torch.randn(32, 3, 30, 112, 112, device=dev)
for _ in range(6):
    y = net(x)
torch.cuda.synchronize()
print(float(y.float().abs().mean()))
