"""fp32 conv-GEMM shapes of the training step (batch 4 x 2^18 samples, exp/train_diffusion_gh.yaml) through sf_bench_conv1d:
python tools/gemm_f32.py [path] [tile]   (path 0 = the dispatcher's choice, 6 = macro tile)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from syncfusion_amd import _lib

lib = _lib.load()
torch.zeros(1, device="cuda")
path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tile = int(sys.argv[2]) if len(sys.argv) > 2 else -1
SHAPES = [  # B, L, C, N, taps
    (4, 4096, 128, 128, 3), (4, 2048, 256, 256, 3), (4, 1024, 512, 512, 3), (4, 512, 512, 512, 3), (4, 256, 1024, 1024, 3),
    (4, 2048, 256, 1536, 1), (4, 1024, 512, 1536, 1), (4, 256, 1024, 1536, 1), (4, 2048, 512, 256, 1), (4, 256, 512, 1024, 1),
    (8, 352, 256, 256, 3), (8, 176, 512, 512, 3), (8, 44, 1024, 1024, 3),
]
for B, L, Cc, N, taps in SHAPES:
    ms = C.c_float()
    rc = lib.sf_bench_conv1d(0, B, L, Cc, N, taps, 1, path, tile, -1, 20, C.byref(ms))
    gf = 2.0 * B * L * N * taps * Cc / 1e9
    print(f"B={B} L={L} C={Cc} N={N} taps={taps}: rc {rc} {ms.value * 1e3:8.1f} us  {gf / ms.value:7.1f} TFLOP/s" if rc == 0 else f"B={B} L={L} C={Cc} N={N} taps={taps}: rc {rc}")
