"""Macro-tile variants on the guidance-batch GEMM shapes of the U-Net (64 evaluations):  python tools/mt_variants.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [("d3 conv3", 64, 704, 128, 128, 3), ("d4 conv3", 64, 352, 256, 256, 3), ("d5 conv3", 64, 176, 512, 512, 3),
          ("d6 conv3", 64, 88, 1024, 1024, 3), ("d7 conv3", 64, 44, 1024, 1024, 3), ("d4 qkv", 64, 352, 256, 1536, 1),
          ("d5 qkv", 64, 176, 512, 1536, 1), ("d6 out", 64, 88, 512, 1024, 1), ("d5 1x1", 64, 176, 512, 512, 1),
          ("ref d4 conv3", 20, 2048, 256, 256, 3), ("ref d5 conv3", 20, 1024, 512, 512, 3), ("ref d6 conv3", 20, 512, 1024, 1024, 3)]
for name, B, L, Cc, N, taps in shapes:
    row = []
    for v in (0, 1, 3, 5):
        env = dict(os.environ, SF_MT_VARIANT=str(v))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_one.py"), str(B), str(L), str(Cc), str(N), str(taps), "6", "-1", "-1", "50"],
                             capture_output=True, text=True, env=env).stdout
        us = float(out.strip().split()[-1])
        row.append(us)
    fl = 2.0 * B * L * N * Cc * taps
    print(f"{name:14s} M={B*L:6d} N={N:5d} K={Cc*taps:5d}  " + "  ".join(f"v{v}:{u:7.1f}us({fl/u/1e6:5.0f}TF)" for v, u in zip((0, 1, 3, 5), row)), flush=True)
