"""Tuning aid: conv_gemm_v2 / classic tiles on MFMA-bound shapes (CFG batch, onset net) -- python tools/gemm_big.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from syncfusion_amd import _lib
lib = _lib.load(); torch.zeros(1, device="cuda")
shapes = [("big square", 64, 704, 512, 512, 3), ("d6 conv3 B64", 64, 88, 1024, 1024, 3), ("d4 conv3 B64", 64, 352, 256, 256, 3),
          ("onset-like 128ch", 8, 11264, 128, 256, 9), ("wide 1x1", 64, 704, 1024, 1536, 1), ("huge", 64, 2816, 256, 256, 3)]
for name, B, L, Cc, N, taps in shapes:
    fl = 2.0 * B * L * N * taps * Cc
    row = []
    for vn, path, tile in (("auto", 0, -1), ("v2 128x128", 4, 0), ("v2 128x64", 4, 1), ("v2 64x64", 4, 2)):
        ms = C.c_float()
        rc = lib.sf_bench_conv1d(1, B, L, Cc, N, taps, 1, path, tile, -1, 30, C.byref(ms))
        row.append(f"{vn}={ms.value*1e3:.0f}us({fl/ms.value/1e9:.0f}TF)" if rc == 0 else f"{vn}=n/a")
    print(f"  {name:18s} M={B*L} N={N} K={taps*Cc}: " + "  ".join(row), flush=True)
