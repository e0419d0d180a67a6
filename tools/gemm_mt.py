"""Macro-tile kernel (conv_gemm_mt, each tile variant) vs conv_gemm_v2 on the MFMA-bound shapes (python tools/gemm_mt.py [bf16|fp16])."""
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
dt = {"bf16": 1, "fp16": 2}[sys.argv[1]]
paths = eval(sys.argv[2])
def run(B, L, Cc, N, taps, path, tile, iters=30):
    ms = C.c_float()
    rc = lib.sf_bench_conv1d(dt, B, L, Cc, N, taps, 1, path, tile, 1 if path == 4 else -1, iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else None
shapes = [  # (name, B, L, C, N, taps): U-Net levels at 64 evaluations (configs[2]) and onset-net-like long activations
    ("d3 conv3 x64", 64, 704, 128, 128, 3), ("d4 conv3 x64", 64, 352, 256, 256, 3), ("d5 conv3 x64", 64, 176, 512, 512, 3),
    ("d6 conv3 x64", 64, 88, 1024, 1024, 3), ("d7 conv3 x64", 64, 44, 1024, 1024, 3), ("d7 conv3 x32", 32, 44, 1024, 1024, 3),
    ("d7 qkv x64", 64, 44, 1024, 1536, 1), ("d5 qkv x64", 64, 176, 512, 1536, 1), ("d6 out x64", 64, 88, 512, 1024, 1),
    ("long 128->256 k9", 8, 11264, 128, 256, 9), ("long 64->192 k9", 32, 11264, 64, 192, 9), ("long 192->64 k3", 32, 11264, 192, 64, 3),
    ("long 256->576 k9", 8, 5632, 256, 576, 9), ("long 512->960 k9", 8, 1408, 512, 960, 9), ("square 8k", 8, 1024, 4096, 4096, 1),
]
for name, B, L, Cc, N, taps in shapes:
    fl = 2.0 * B * L * N * taps * Cc
    row = []
    for vn, path, tile in paths:
        us = run(B, L, Cc, N, taps, path, tile)
        row.append(f"{vn}={us:.1f}us({fl / us / 1e6:.0f}TF)" if us else f"{vn}=n/a")
    print(f"{name} [M={B*L} N={N} K={taps*Cc}]: " + "  ".join(row), flush=True)
''' % HERE
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for variant, label in (("0", "mt 256x128"), ("1", "mt 128x128"), ("2", "mt 128x192")):
    print(f"== SF_MT_VARIANT={variant}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD, dtype, repr([(label, 6, -1)])], env=dict(os.environ, SF_MT_VARIANT=variant), check=False)
print("== automatic choice and conv_gemm_v2", flush=True)
subprocess.run([sys.executable, "-c", CHILD, dtype, repr([("auto", 0, -1), ("v2 128x64", 4, 1), ("v2 64x64", 4, 2)])], check=False)
