#!/usr/bin/env python3
"""Benchmark of the SyncFusion denoising hot path on MI355X (driver contract: see the task brief).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): batch 8 clips per GPU, 2 s @ 22.05 kHz -> L0 = 45056 (= 44 * 1024, the
nearest legal U-Net length >= 44100; SURVEY.md finding 4), bf16, embedding_scale 1.0 (one U-Net evaluation per
denoise step), dummy conditioning (zero CLAP embedding, onset track with one impulse at sample 0), random-init
215 M-parameter U-Net under manual_seed(1234).  A "step" is one iteration of the v-sampler loop for the whole
batch: one U-Net evaluation + the sampler update, exactly what DiffusionModel.sample runs per step.

Timed region: `sample(num_steps=K)` -- inputs resident in HBM, barrier + synchronize on both sides, max over
ranks.  value = N * K / t  (batch-steps per second summed over the N independent per-GPU batches; weak scaling).

Extra objects on the JSON line:
  roofline      the dominant kernel (largest share of a step's device time).  `achieved` = the ALGORITHMIC FLOPs
                of its launches in one evaluation / their summed durations, measured with HIP events recorded
                on the launch stream around every kernel of an instrumented evaluation in this process.
  cpu_baseline  the CPU oracle (oracle/unet_ref.py, a port -- the reference's U-Net source is not in its tree)
                timed on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

L0 = 45056
BATCH = 8
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def build_model(dtype: str, device):
    from syncfusion_amd.reference_config import model_config
    import syncfusion_amd as sa

    import contextlib

    torch.manual_seed(1234)
    cfg = model_config()
    cfg["model"]["net_t"]["dtype"] = dtype
    with contextlib.redirect_stdout(sys.stderr):   # the Model constructor prints like the reference's; stdout carries ONE JSON line
        model = sa.instantiate(cfg)
    return model.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--scale", type=float, default=1.0, help="embedding_scale (!= 1 -> classifier-free guidance, 2 evals/step)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch event timings of one evaluation to this file")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the N > 1 code path on a 1-GPU box")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from syncfusion_amd import dist as sfd

    sfd.FORCE_COLLECTIVES = bool(args.force_dist)
    import syncfusion_amd as sa

    B = args.batch
    model = build_model(args.dtype, device)
    model.model.sampler.use_graph = not args.no_graph
    bcast_bytes = sfd.broadcast_module(model, src=0)          # RCCL broadcast of the weights, once, outside the timed loop
    net = model.model.net

    # synthetic inputs, identical bits on every run: per-rank noise seed 1000 + rank
    noise = torch.randn(B, 1, L0, generator=torch.Generator().manual_seed(sfd.rank_seed(1000, rank))).to(device)
    track = torch.zeros(B, 1, L0, device=device)
    track[:, 0, 0] = 1.0                                       # "dummy cond": one impulse at sample 0
    _, info = model.onsets_encoder(track, with_info=True)
    channels = info["xs"][2:-1]
    emb = torch.zeros(B, 1, 512, device=device)                # dummy CLAP embedding

    def run(steps):
        return model.model.sample(x_noisy=noise, num_steps=steps, channels=channels, embedding=emb, embedding_scale=args.scale)

    def fence():
        torch.cuda.synchronize(device)
        if dist_on:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize(device)

    if not args.no_graph:
        run(2)                                                 # setup, untimed: instantiates the step graphs (reused by every later call)
    if args.warmup > 0:
        run(args.warmup)
    fence()
    t0 = time.perf_counter()
    out = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist

        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    assert torch.isfinite(out).all()
    gathered = sfd.gather_clips(out, B * world, dst=0)         # RCCL gather of the finished clips, once, after the loop

    if rank != 0:
        if world > 1:
            import torch.distributed as dist

            dist.destroy_process_group()
        return

    # ---------------- roofline: per-kernel HIP-event timing of one instrumented evaluation ----------------
    evals = 2 if args.scale != 1.0 else 1
    sigma = torch.full((B,), 0.5, device=device)
    net.engine().profile_forward(noise, sigma, channels, emb, args.scale)          # warm
    recs = net.engine().profile_forward(noise, sigma, channels, emb, args.scale)
    if args.dump_launches:
        with open(args.dump_launches, "w") as f:
            for i, (label, ms, fl, by) in enumerate(recs):
                f.write(f"{i}\t{label}\t{ms * 1e3:.2f}us\t{fl / 1e6:.2f}MFLOP\t{by / 1e6:.3f}MB\n")
    # HIP events bracket launch latency as well as execution: the engine times an empty kernel the same way
    # ("calib_empty"); its duration minus ~1.5 us of real execution is the fixed cost subtracted from every launch
    calib = sorted(ms for label, ms, _, _ in recs if label == "calib_empty")
    # Calibrated against rocprofv3 --kernel-trace of the same command (profiles/r1_*_kernel_stats.csv): about half of the
    # empty-kernel event time overlaps with a real kernel's own launch ramp, so half of it is subtracted.
    overhead_ms = 0.5 * max(0.0, (calib[len(calib) // 2] if calib else 0.0) - 1.5e-3)
    recs = [(label, max(ms - overhead_ms, 1e-4), fl, by) for label, ms, fl, by in recs if label != "calib_empty"]
    agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for label, ms, fl, by in recs:
        a = agg[label]
        a[0] += ms
        a[1] += fl
        a[2] += by
        a[3] += 1
    total_ms = sum(a[0] for a in agg.values())
    dom = max(agg, key=lambda k: agg[k][0])
    d_ms, d_fl, d_by, d_n = agg[dom]
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    # the roofline that bounds the dominant kernel: whichever of its algorithmic FLOPs / MFMA peak and algorithmic bytes / HBM
    # peak takes longer (the small-batch GEMMs sit BELOW the machine balance of ~310 FLOP/B: they are HBM-side kernels)
    mfma_kernel = d_fl / (peak * 1e12) >= d_by / (PEAK_HBM_GBS * 1e9)
    if mfma_kernel:
        achieved = d_fl / (d_ms * 1e-3) / 1e12
        roof = dict(bound="mfma", achieved=round(achieved, 3), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 5))
    else:
        achieved = d_by / (d_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", achieved=round(achieved, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(achieved / PEAK_HBM_GBS, 5))
    # HBM-side bytes per launch of the dominant kernel: PMC passes cannot run inside this process (they serialise the
    # device and need rocprofv3), so the number comes from the committed summary of separate rocprofv3 --pmc passes over
    # THIS command (tools/pmc_traffic.py; FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is); null when the
    # summary has no entry for the kernel or the workload is not the profiled one.
    traffic, traffic_src = None, None
    if args.batch == 8 and args.dtype == "bf16" and args.scale == 1.0:
        import glob

        for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))[::-1]:
            try:
                with open(path) as f:
                    kern = json.load(f).get("kernels", {})
            except (OSError, ValueError):
                continue
            ent = kern.get(dom.split("<")[0])
            if ent:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
                break
    roof.update(traffic=traffic, traffic_source=traffic_src, algorithmic_bytes_per_launch=round(d_by / d_n), kernel=dom, launches_per_eval=d_n, avg_launch_us=round(d_ms / d_n * 1e3, 3),
                share_of_eval_time=round(d_ms / total_ms, 4),
                per_kernel={k: dict(ms=round(v[0], 4), launches=v[3], tflops=round(v[1] / max(v[0], 1e-9) / 1e9, 3),
                                    gbs=round(v[2] / max(v[0], 1e-9) / 1e6, 1)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])},
                eval_device_ms=round(total_ms, 4), launches_per_eval_total=len(recs),
                event_overhead_us_subtracted=round(overhead_ms * 1e3, 3))
    # whole-step roofline: t_roofline = max(alg FLOPs / MFMA peak, alg bytes / HBM peak) summed per launch
    t_roof = sum(max(fl / (peak * 1e12), by / (PEAK_HBM_GBS * 1e9)) for _, _, fl, by in recs) * 1e3
    ms_per_step = elapsed / args.steps * 1e3
    roof["step_roofline_ms"] = round(t_roof, 4)
    roof["step_roofline_frac"] = round(t_roof / (ms_per_step / 1.0), 5)

    # ---------------- cpu_baseline: the oracle on this box's host cores, bounded sample ----------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import unet_ref          # the checker, timed here as the CPU baseline -- never on the product path

        # host cores this process may actually use (cgroup/affinity), capped: oversubscribing a 256-thread box
        # with torch's intra-op pool makes the small-channel conv1d calls of the oracle orders of magnitude slower
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(ncores, 32)))
        P = {"net." + k: v.detach().float().cpu() for k, v in net.state_dict().items()}
        cfg = dict(net.hparams)
        xc, ec = noise.cpu(), emb.cpu()
        cc = [c.cpu() for c in channels]
        sc = torch.full((B,), 0.5)
        with torch.no_grad():
            t1 = time.perf_counter()
            unet_ref.unet_forward(P, cfg, xc, sc, embedding=ec, channels=cc, embedding_scale=args.scale)   # warm-up step
            one = time.perf_counter() - t1
            n = max(1, min(50, int(args.cpu_seconds / max(one, 1e-3)) - 1))
            t1 = time.perf_counter()
            for _ in range(n):
                unet_ref.unet_forward(P, cfg, xc, sc, embedding=ec, channels=cc, embedding_scale=args.scale)
            dt = time.perf_counter() - t1
        cpu = dict(value=round(n / dt, 4), unit="steps/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"{n} timed denoise steps (U-Net evaluation, batch {B}, L0 {L0}, fp32) after 1 warm-up; "
                          "the sampler's element-wise update is excluded (<0.01% of a step)")

    line = {
        "metric": "U-Net denoise steps/sec (batch x 2 s@22.05 kHz)",
        "value": round(world * args.steps / elapsed, 3),
        "unit": "steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"batch={B}/GPU, {args.steps}-step DDIM (v-sampler), L0={L0} (2 s @ 22.05 kHz padded to 44*1024), "
                               f"{args.dtype}, embedding_scale={args.scale} ({evals} U-Net eval/step), dummy cond; BASELINE configs[1]",
                   "batch_per_gpu": B, "L0": L0, "evals_per_step": evals, "clip_steps_per_s": round(world * B * args.steps / elapsed, 2),
                   "params_M": round(sum(p.numel() for p in net.parameters()) / 1e6, 2), "hip_graph": not args.no_graph,
                   "weights_broadcast_bytes": bcast_bytes, "gathered_clips": None if gathered is None else int(gathered.shape[0])},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if dist_on:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
