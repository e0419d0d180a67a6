import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from syncfusion_amd import VideoOnsetNet
from oracle.onsetnet_ref import onsetnet_flops
dev = torch.device('cuda:0')
for dtype in ('bf16', 'fp32'):
    for N in (1, 8, 32):
        torch.manual_seed(0)
        net = VideoOnsetNet(False, dtype=dtype).to(dev).eval()
        x = torch.randn(N, 3, 30, 112, 112, device=dev)
        y = net(x); torch.cuda.synchronize()
        t = time.perf_counter(); reps = 5
        for _ in range(reps): y = net(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        print(f"onset {dtype} N={N}: {dt*1e3:.2f} ms/forward, {N/dt:.1f} clips/s, {onsetnet_flops(30,112,112)*N/dt/1e12:.1f} TFLOP/s", flush=True)
