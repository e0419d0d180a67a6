"""Fixed vs per-K cost of the short-activation GEMM kernels: M = 176 rows (4 clips x 44), N = 1024, K swept (python tools/gemm_kscale.py)."""
import ctypes as C
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
for B, L in ((4, 44), (8, 44), (4, 352)):
    for path, tile, name in ((5, 2, "wp32"), (2, 2, "fast32")):
        row = []
        for Cc in (64, 128, 256, 512, 1024, 2048, 3072):
            ms = C.c_float()
            rc = lib.sf_bench_conv1d(1, B, L, Cc, 1024, 1, 1, path, tile, -1, 400, C.byref(ms))
            row.append(f"K={Cc}:{ms.value*1e3:.2f}" if rc == 0 else f"K={Cc}:n/a")
        print(f"  M={B*L:5d} {name:7s} " + "  ".join(row), flush=True)
''' % HERE
for cold in ("0", "1"):
    print(f"COLD={cold}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SF_BENCH_COLD=cold), check=False)
