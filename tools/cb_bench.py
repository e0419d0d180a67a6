"""Tuning aid (python tools/cb_bench.py): the channel-block split-K chain of conv_cb.hip against the launches it replaces, alone on
the chip, warm and HBM-cold weights, per U-Net depth at four clips per branch (configs[1]) and at larger batches.  One process per
setting (the SF_CB_MT / SF_BENCH_COLD knobs are read once)."""
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
cold = int(os.environ.get("SF_BENCH_COLD", "0"))
shapes = [("d7", 4, 44, 1024), ("d6", 4, 88, 1024), ("d5", 4, 176, 512), ("d4", 4, 352, 256), ("d3", 4, 704, 128),
          ("d7 B16", 16, 44, 1024), ("d6 B16", 16, 88, 1024), ("d5 B16", 16, 176, 512), ("d4 B16", 16, 352, 256),
          ("d7 B32", 32, 44, 1024), ("d6 B32", 32, 88, 1024)]
for name, B, L, Cc in shapes:
    ms = (C.c_float * 4)()
    rc = lib.sf_bench_conv_cb(1, B, L, Cc, 8, int(os.environ.get("SF_CB_KB", "1")), cold, 200, ms)
    old = C.c_float()
    rc2 = lib.sf_bench_conv1d(1, B, L, Cc, Cc, 3, 1, 0, -1, -1, 200, C.byref(old))
    cb = "conv %%.1f  reduce_gn %%.1f  conv+gn %%.1f  reduce_ln %%.1f us" %% tuple(v * 1e3 for v in ms) if rc == 0 else "n/a"
    print(f"  {name:8s} M={B*L:5d} C={Cc:4d}  cb: {cb}   | launch_conv_gemm auto: {old.value*1e3:.1f} us" if rc2 == 0 else f"  {name} cb: {cb}", flush=True)
''' % HERE
for cold in ("0", "1"):
    for mt in ("0", "1", "2", "3", "4"):
        print(f"COLD={cold} SF_CB_MT={mt} (0 = automatic)", flush=True)
        env = dict(os.environ, SF_BENCH_COLD=cold, SF_CB_MT=mt)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
