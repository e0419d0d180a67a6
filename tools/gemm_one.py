"""Run ONE conv-GEMM shape / variant many times (for rocprofv3 --pmc):  python tools/gemm_one.py B L C N taps path tile sk [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from syncfusion_amd import _lib

lib = _lib.load()
torch.zeros(1, device="cuda")
B, L, Cc, N, taps, path, tile, sk = [int(v) for v in sys.argv[1:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
ms = C.c_float()
rc = lib.sf_bench_conv1d(1, B, L, Cc, N, taps, 1, path, tile, sk, iters, C.byref(ms))
print("rc", rc, "us", ms.value * 1e3)
