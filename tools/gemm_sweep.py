"""Tuning aid: time the conv-GEMM kernel families on the U-Net's layer shapes (python tools/gemm_sweep.py)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime the library binds to)
from syncfusion_amd import _lib

lib = _lib.load()
torch.zeros(1, device="cuda")

def run(dt, B, L, Cc, N, taps, up, path, tile, sk, iters=50):
    ms = C.c_float()
    rc = lib.sf_bench_conv1d(dt, B, L, Cc, N, taps, up, path, tile, sk, iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else None

shapes = [  # (name, B, L, C, N, taps)
    ("onset-like 128ch", 8, 11264, 128, 256, 9), ("big square", 64, 704, 512, 512, 3),
    ("d7 conv3 B8", 8, 44, 1024, 1024, 3), ("d6 conv3 B8", 8, 88, 1024, 1024, 3), ("d5 conv3 B8", 8, 176, 512, 512, 3),
    ("d4 conv3 B8", 8, 352, 256, 256, 3), ("d3 conv3 B8", 8, 704, 128, 128, 3), ("d6 qkv B8", 8, 88, 1024, 1536, 1),
    ("d6 conv3 B32", 32, 88, 1024, 1024, 3), ("d4 conv3 B32", 32, 352, 256, 256, 3), ("d6 conv3 B64", 64, 88, 1024, 1024, 3),
]
variants = [("auto", 0, -1, -1),
            ("wsk 64x64", 2, 0, -1), ("wsk 64x32", 2, 1, -1), ("wsk 32x32", 2, 2, -1),
            ("wp 64x64", 5, 0, -1), ("wp 64x32", 5, 1, -1), ("wp 32x32", 5, 2, -1),
            ("v2 128x128", 4, 0, 1), ("v2 128x64", 4, 1, 1), ("v2 64x64", 4, 2, 1)]
dt = 1 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else 0
for name, B, L, Cc, N, taps in shapes:
    fl = 2.0 * B * L * N * taps * Cc
    row = []
    for vn, path, tile, sk in variants:
        us = run(dt, B, L, Cc, N, taps, 1, path, tile, sk)
        row.append(f"{vn}={us:.1f}us({fl / us / 1e6:.0f}TF)" if us else f"{vn}=n/a")
    print(f"{name} [M={B*L} N={N} K={taps*Cc}]: " + "  ".join(row), flush=True)
