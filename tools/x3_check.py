#!/usr/bin/env python3
"""fp32x (split fp16 operands) against fp32 and fp64: op level through sf_op_conv1d_cl, engine level through sample().
    python3 tools/x3_check.py [ops] [engine] [time]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F

from syncfusion_amd import _lib

cuda = torch.device("cuda:0")
lib = _lib.load()
what = set(sys.argv[1:]) or {"ops", "engine", "time"}


def conv_case(dtype, B, L, C, N, taps, up=1, residual=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, L, generator=g) * 1.5 + 0.3
    w = torch.randn(N, C, taps, generator=g) / (C * taps) ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    h = x.double()
    if up > 1:
        h = F.interpolate(h, scale_factor=up, mode="nearest")
    ref = F.conv1d(h, w.double(), bias.double(), padding=taps // 2)
    Lout = ref.shape[-1]
    res = torch.randn(B, N, Lout, generator=g) if residual else None
    if residual:
        ref = ref + res.double()
    x_cl = x.transpose(1, 2).contiguous().to(cuda)
    res_cl = res.transpose(1, 2).contiguous().to(cuda) if residual else None
    out = torch.empty(B, Lout, N, device=cuda)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=cuda)
    wd, bd = w.to(cuda), bias.to(cuda)
    rc = lib.sf_op_conv1d_cl(_lib.DTYPES[dtype], x_cl.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, None, 0, 1e-5,
                             res_cl.data_ptr() if residual else None, B, L, C, N, taps, 1, taps // 2, up, out.data_ptr(), ws.data_ptr(), ws.numel(),
                             _lib.stream_ptr(cuda))
    _lib.check(rc, "sf_op_conv1d_cl")
    torch.cuda.synchronize()
    got = out.double().cpu().transpose(1, 2)
    return float((got - ref).norm() / ref.norm())


if "ops" in what:
    shapes = [(8, 5632, 64, 128, 3, 1), (9, 5000, 128, 320, 1, 1), (4, 4096, 128, 128, 3, 1), (3, 3000, 256, 192, 1, 1), (8, 8192, 128, 128, 3, 1),
              (8, 1408, 512, 512, 3, 1), (32, 176, 1024, 1024, 3, 1), (32, 176, 1024, 1536, 1, 1), (8, 2816, 64, 64, 3, 2),
              (4, 88, 1024, 1024, 3, 1), (8, 44, 512, 256, 3, 1), (2, 100, 256, 320, 1, 1), (2, 176, 128, 256, 1, 1), (4, 352, 256, 256, 3, 1)]
    for sh in shapes:
        B, L, C, N, taps, up = sh
        e32 = conv_case("fp32", B, L, C, N, taps, up)
        ex = conv_case("fp32x", B, L, C, N, taps, up)
        print(f"conv {sh}: rel-L2 vs fp64  fp32 {e32:.2e}  fp32x {ex:.2e}", flush=True)

if "time" in what:
    import ctypes as C

    for (B, L, Cc, N, taps) in [(32, 176, 1024, 1024, 3), (32, 176, 1024, 1536, 1), (32, 352, 512, 512, 3), (32, 704, 256, 256, 3), (32, 1408, 128, 128, 3),
                                (64, 44, 1024, 1024, 3), (4, 4096, 512, 512, 3), (4, 16384, 128, 128, 3), (4, 44, 1024, 1024, 3), (4, 176, 512, 512, 3)]:
        row = []
        for dt in ("fp32", "fp32x", "bf16"):
            ms = C.c_float()
            rc = lib.sf_bench_conv1d(_lib.DTYPES[dt], B, L, Cc, N, taps, 1, 0, -1, -1, 50, C.byref(ms))
            row.append(f"{dt} {ms.value * 1e3:8.1f} us" if rc == 0 else f"{dt} n/a")
        fl = 2.0 * B * L * N * taps * Cc
        print(f"gemm M={B * L} N={N} K={taps * Cc}: " + "  ".join(row) + f"   ({fl / 1e9:.1f} GFLOP)", flush=True)

if "engine" in what:
    sys.argv = sys.argv[:1]
    import bench

    model = bench.build_model("fp32", cuda)
    net = model.model.net
    for (B, scale, steps) in [(8, 1.0, 10), (32, 2.0, 6)]:
        nz = torch.randn(B, 1, bench.L0, generator=torch.Generator().manual_seed(1000)).to(cuda)
        ch, e = bench.synthetic_conditioning(model, B, bench.L0, cuda, real=True)
        outs, rates = {}, {}
        for dt in ("fp32", "fp32x"):
            net.compute_dtype = dt
            model.model.sample(x_noisy=nz, num_steps=2, channels=ch, embedding=e, embedding_scale=scale)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs[dt] = model.model.sample(x_noisy=nz, num_steps=steps, channels=ch, embedding=e, embedding_scale=scale)
            torch.cuda.synchronize()
            rates[dt] = steps / (time.perf_counter() - t0)
            sig = torch.full((B,), 0.5, device=cuda)
            net.engine().profile_forward(nz, sig, ch, e, scale)
            recs = net.engine().profile_forward(nz, sig, ch, e, scale)
            agg = {}
            for label, ms, fl, by in recs:
                a = agg.setdefault(label, [0.0, 0])
                a[0] += ms
                a[1] += 1
            print(f"  [{dt} B={B} scale={scale}] " + ", ".join(f"{k} {v[0]:.3f}ms/{v[1]}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]))
        rel = float((outs["fp32x"].double() - outs["fp32"].double()).norm() / outs["fp32"].double().norm())
        print(f"engine B={B} scale={scale} steps={steps}: fp32 {rates['fp32']:.1f} steps/s, fp32x {rates['fp32x']:.1f} steps/s, final-sample rel-L2 {rel:.2e}", flush=True)
