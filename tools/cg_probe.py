"""How does the wave-private GEMM's time depend on the NUMBER of workgroups and on the bytes each one streams?
(python tools/cg_probe.py)  K = 3072 unless stated; every workgroup of a bm x bn tile streams (bm + bn) * K * 2 bytes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(%r))
import torch
from syncfusion_amd import _lib
lib = _lib.load(); torch.zeros(1, device="cuda")
def run(B, L, K, N, tile, path=5):
    ms = C.c_float()
    rc = lib.sf_bench_conv1d(1, B, L, K, N, 1, 1, path, tile, -1, 300, C.byref(ms))
    return ms.value * 1e3 / 1.0 if rc == 0 else float("nan")
print("32x32 tiles, K=3072, one row tile (M=32), WGs = N/32:")
for N in (256, 1024, 2048, 4096, 6144, 8192, 12288, 16384):
    print(f"   WGs={N//32:4d}: {run(1, 32, 3072, N, 2):7.2f} us", flush=True)
print("32x32 tiles, K=3072, M=176 (6 row tiles), N swept:")
for N in (128, 256, 512, 1024, 2048):
    print(f"   WGs={6*N//32:4d}: {run(4, 44, 3072, N, 2):7.2f} us", flush=True)
print("64x64 tiles, K=3072: few big workgroups (786 KB each)")
for B, L, N in ((1, 64, 512), (4, 64, 512), (4, 64, 1024), (4, 64, 2048), (8, 64, 2048)):
    print(f"   WGs={(B*L//64)*(N//64):4d}: {run(B, L, 3072, N, 0):7.2f} us", flush=True)
print("64x32 tiles, K=3072:")
for B, L, N in ((1, 64, 1024), (4, 64, 1024), (4, 64, 2048)):
    print(f"   WGs={(B*L//64)*(N//32):4d}: {run(B, L, 3072, N, 1):7.2f} us", flush=True)
print("K sweep, 32x32, M=176, N=1024:")
for K in (64, 256, 512, 1024, 1536, 2048, 3072):
    print(f"   K={K:5d}: wp {run(4, 44, K, 1024, 2):7.2f} us   fast {run(4, 44, K, 1024, 2, path=2):7.2f} us", flush=True)
''' % HERE
for cold in ("0", "1"):
    print(f"COLD={cold}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SF_BENCH_COLD=cold), check=False)
