"""Tuning aid (python tools/rs_bench.py): the register-staged small-batch GEMM (conv_gemm_rs.hip) against the wave-private kernel it
replaces, alone on the chip, on the 1x1 / Linear shapes of the deep levels at four clips per branch; warm and HBM-cold weights.
One process per setting (SF_BENCH_COLD / SF_BENCH_NO_WFR are read once)."""
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
shapes = [("d7 inject(K1280)", 4, 44, 1280, 1024, 1), ("d6 inject", 4, 88, 1280, 1024, 1), ("d5 inject(K640)", 4, 176, 640, 512, 1), ("d4 inject(K320)", 4, 352, 320, 256, 1),
          ("d7 qkv", 4, 44, 1024, 1536, 1), ("d6 qkv", 4, 88, 1024, 1536, 1), ("d5 qkv", 4, 176, 512, 1536, 1), ("d4 qkv", 4, 352, 256, 1536, 1),
          ("d7 out", 4, 44, 512, 1024, 1), ("d6 out", 4, 88, 512, 1024, 1), ("d5 out", 4, 176, 512, 512, 1), ("d4 out", 4, 352, 512, 256, 1),
          ("d6 down(K1024)", 4, 88, 1024, 1024, 1), ("d5 down", 4, 176, 512, 512, 1), ("d4 up conv3 (K1536)", 4, 352, 512, 256, 3), ("d3 conv3 (K384)", 4, 704, 128, 128, 3)]
for name, B, L, Cc, N, taps in shapes:
    ms = C.c_float()
    rc = lib.sf_bench_conv1d(1, B, L, Cc, N, taps, 1, 0, -1, -1, 300, C.byref(ms))
    print(f"  {name:22s} M={B*L:5d} K={taps*Cc:5d} N={N:5d}  {ms.value*1e3:6.2f} us" if rc == 0 else f"  {name} n/a", flush=True)
''' % HERE
for cold in ("0", "1"):
    for nowfr in ("", "1"):
        print(f"COLD={cold}  {'wave-private (no fragment-ordered weights)' if nowfr else 'register-staged where eligible'}", flush=True)
        env = dict(os.environ, SF_BENCH_COLD=cold)
        if nowfr:
            env["SF_BENCH_NO_WFR"] = "1"
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
