"""Tuning aid: the deep-level GEMM shapes of the U-Net with WARM vs COLD (HBM-streamed) weights and the
prefetch depth of conv_gemm_fast (python tools/gemm_cold.py).  Spawns one process per setting (the knobs are read once)."""
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
shapes = [("d7 conv3", 8, 44, 1024, 1024, 3), ("d6 conv3", 8, 88, 1024, 1024, 3), ("d5 conv3", 8, 176, 512, 512, 3),
          ("d4 conv3", 8, 352, 256, 256, 3), ("d3 conv3", 8, 704, 128, 128, 3), ("d6 qkv", 8, 88, 1024, 1536, 1), ("d7 qkv", 8, 44, 1024, 1536, 1),
          ("d6 out", 8, 88, 512, 1024, 1), ("d4x2 conv3 (B=4)", 4, 352, 256, 256, 3), ("d6 conv3 (B=4)", 4, 88, 1024, 1024, 3)]
for name, B, L, Cc, N, taps in shapes:
    row = []
    for vn, path, tile in (("auto", 0, -1), ("fast32", 2, 2), ("fast64x32", 2, 1), ("fast64", 2, 0), ("wp32", 5, 2), ("v2-64", 4, 2)):
        ms = C.c_float()
        rc = lib.sf_bench_conv1d(1, B, L, Cc, N, taps, 1, path, tile, 1 if path == 4 else -1, 200, C.byref(ms))
        row.append(f"{vn}={ms.value*1e3:.1f}" if rc == 0 else f"{vn}=n/a")
    print(f"  {name:18s} " + "  ".join(row), flush=True)
''' % HERE
for cold in ("0", "1"):
    for depth in ("2", "4", "6"):
        print(f"COLD={cold} FAST_DEPTH={depth}", flush=True)
        env = dict(os.environ, SF_BENCH_COLD=cold, SF_FAST_DEPTH=depth)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
