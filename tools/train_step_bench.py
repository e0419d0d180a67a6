#!/usr/bin/env python3
"""Time one training step of the reference's configuration (exp/train_diffusion_gh.yaml: fp32, batch 4 per device, clips of
2^18 samples) on the HIP training path: Model.training_step -> loss.backward() -> AdamW.step().

    python tools/train_step_bench.py [--batch 4] [--length 262144] [--steps 3] [--no-optimizer]
Prints one JSON line: ms per step (forward / backward / optimizer), clips/s, peak HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def graphed(args, torch, model, opt, batch) -> int:
    """Forward + backward captured once; every step = seed the noise on the side stream, replay, optimizer step."""
    params = [p for p in model.parameters() if p.requires_grad]
    x, y = batch[0], batch[1]
    sig = torch.rand(x.shape[0], device=x.device)
    noise = torch.randn_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # warm-up on the capture stream (allocator pools, lazy kernel loads)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            z = model.clap_encode_audio(x)
            _, info = model.onsets_encoder(y, with_info=True)
            loss = model.model(x, channels=info["xs"][2:-1], embedding=z, sigmas=sig, noise=noise)
            loss.backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        z = model.clap_encode_audio(x)
        _, info = model.onsets_encoder(y, with_info=True)
        static_loss = model.model(x, channels=info["xs"][2:-1], embedding=z, sigmas=sig, noise=noise)
        static_loss.backward()
    losses, t_g, t_o = [], 0.0, 0.0
    for it in range(args.warmup + args.steps):
        sig.uniform_()
        noise.normal_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if not args.no_optimizer:
            opt.step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        losses.append(float(static_loss))
        if it >= args.warmup:
            t_g += t1 - t0
            t_o += t2 - t1
    n = args.steps
    total = (t_g + t_o) / n
    print(json.dumps({"workload": f"training step fp32 (graph replay), batch {args.batch}, L0 {args.length}", "fwd_bwd_ms": 1e3 * t_g / n,
                      "optimizer_ms": 1e3 * t_o / n, "step_ms": 1e3 * total, "clips_per_s": args.batch / total,
                      "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9, "losses": losses}))
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--length", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--fused", action="store_true", help="torch.optim.AdamW(fused=True) with the reference's hyper-parameters")
    ap.add_argument("--graph", action="store_true", help="capture forward + backward in a torch CUDA graph (static batch buffers) and replay it")
    args = ap.parse_args()
    import torch

    import syncfusion_amd as sa
    from syncfusion_amd.reference_config import model_config

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = sa.instantiate(model_config()).to(dev)
    opt = model.configure_optimizers()
    if args.fused:
        opt = torch.optim.AdamW(list(model.model.parameters()) + list(model.onsets_encoder.parameters()), lr=model.lr, betas=(model.lr_beta1, model.lr_beta2),
                                eps=model.lr_eps, weight_decay=model.lr_weight_decay, fused=True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.batch, 1, args.length, generator=g).to(dev)
    y = (torch.rand(args.batch, 1, args.length, generator=g) < 0.0005).float().to(dev)
    batch = (x, y, x, None, None)
    t_f = t_b = t_o = 0.0
    losses = []
    if args.graph:
        return graphed(args, torch, model, opt, batch)
    for it in range(args.warmup + args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = model.training_step(batch, it)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if not args.no_optimizer:
            opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        losses.append(float(loss.detach()))
        if it >= args.warmup:
            t_f += t1 - t0
            t_b += t2 - t1
            t_o += t3 - t2
    n = args.steps
    total = (t_f + t_b + t_o) / n
    print(json.dumps({"workload": f"training step fp32, batch {args.batch}, L0 {args.length}", "forward_ms": 1e3 * t_f / n, "backward_ms": 1e3 * t_b / n,
                      "optimizer_ms": 1e3 * t_o / n, "step_ms": 1e3 * total, "clips_per_s": args.batch / total,
                      "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9, "losses": losses}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
