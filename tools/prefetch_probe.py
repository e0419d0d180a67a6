"""Would a weight prefetch from idle CUs in the PRECEDING kernel pay?  K = 3072 / 1536 / 1024 GEMMs of the batch-8 step with
HBM-cold weights (SF_BENCH_COLD=1): GEMM alone, touch kernel alone, touch + GEMM (python tools/prefetch_probe.py)."""
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
for B, L, K, N, taps in ((4, 44, 1024, 1024, 3), (4, 88, 1024, 1024, 3), (4, 176, 512, 512, 3), (4, 44, 1024, 1536, 1), (4, 44, 512, 1024, 1)):
    ms = C.c_float()
    rc = lib.sf_bench_conv1d(1, B, L, K, N, taps, 1, 5, 2, -1, 300, C.byref(ms))
    print(f"  M={B*L:4d} N={N:4d} K={taps*K:4d}: {ms.value*1e3:7.2f} us", flush=True)
''' % HERE
for cold in ("0", "1"):
    for pre in ("0", "1064", "64", "1224", "224"):
        print(f"COLD={cold} PREFETCH={pre}", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SF_BENCH_COLD=cold, SF_BENCH_PREFETCH=pre), check=False)
