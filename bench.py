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

`--gpus N` without a torch.distributed environment (no WORLD_SIZE): this process starts N fresh ranks with
`python -m torch.distributed.run` BEFORE touching the GPU and relays rank 0's JSON line.  Under the driver's own
torchrun launch (WORLD_SIZE set) it is a rank; a WORLD_SIZE that disagrees with --gpus is an error.

Extra objects on the JSON line:
  roofline      the dominant kernel (largest share of a step's device time).  `achieved` = the ALGORITHMIC bytes (or
                FLOPs) of its launches in one evaluation / their summed durations, measured with HIP events recorded
                on the launch stream around every kernel of an instrumented evaluation in this process.  The whole-step
                figures (`step_*`) use the closed-form work of syncfusion_amd/workmodel.py (every weight
                counted ONCE per step, independent of how the engine splits the batch).
  cpu_baseline  the CPU oracle (oracle/unet_ref.py, a port -- the reference's U-Net source is not in its tree)
                timed on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
  extra         secondary workloads measured in the same process (rank 0, N = 1 only): the fp32 parity path on
                configs[1], BASELINE configs[2] (batch 32, guidance, real conditioning), the reference's own evaluation
                shape (exp/evaluate_gh_gen.yaml:8,21-23), the onset net at N = 32, and the rel-L2 distance between the
                bf16 and fp32 engines' final samples at the benched shape.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

L0 = 45056
BATCH = 8
BASELINE_STEPS = 50         # BASELINE.json configs[1]: 50-step DDIM
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
ES = {"bf16": 2, "fp16": 2, "fp32": 4, "fp32x": 4}
PEAK_X3_TFLOPS = PEAK_BF16_TFLOPS / 3.0   # fp32x: three fp16 MFMAs per product


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=BASELINE_STEPS)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp32x"])
    ap.add_argument("--scale", type=float, default=1.0, help="embedding_scale (!= 1 -> classifier-free guidance, 2 evals/step)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config0", action="store_true", help="skip cpu_baseline.config0_full (the oracle's whole 10-step guided sample)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads of the `extra` object")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch event timings of one evaluation to this file")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--master-port", type=int, default=29533)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 path (nccl = RCCL; gloo only for the share-GPU smoke run)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="map every local rank onto the visible GPUs round-robin (LOCAL_RANK %% device_count): the whole N-rank launcher -> "
                         "broadcast -> timed loop -> gather -> JSON path on a 1-GPU box.  Throughput of such a run is meaningless.")
    return ap.parse_args(argv)


def spawn_ranks(args, script: str = os.path.abspath(__file__), argv=None) -> int:
    """--gpus N from a plain shell: N fresh child ranks (one per GPU) under torchrun.  The parent never initialises HIP."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(args.master_port), script] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        # rank 0's JSON line, wherever it starts: the ranks share one pipe and the C++ side of torch.distributed (gloo / RCCL banners)
        # writes to it unbuffered, so a partial banner can precede the object on the same line
        at = ln.find('{"metric"')
        obj = None
        if at >= 0:
            try:
                obj, _ = json.JSONDecoder().raw_decode(ln[at:])
            except ValueError:
                obj = None
        if isinstance(obj, dict):
            line = json.dumps(obj)
            if at > 0:
                print(ln[:at], file=sys.stderr)
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank child run failed (rc {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line)
    return 0


def build_model(dtype: str, device, upsample_mode: str = "nearest"):
    import contextlib

    import torch

    import syncfusion_amd as sa
    from syncfusion_amd.reference_config import model_config

    torch.manual_seed(1234)
    cfg = model_config()
    cfg["model"]["net_t"]["dtype"] = dtype
    if upsample_mode != "nearest":
        cfg["model"]["upsample_mode"] = upsample_mode
    with contextlib.redirect_stdout(sys.stderr):   # the Model constructor prints like the reference's; stdout carries ONE JSON line
        model = sa.instantiate(cfg)
    return model.to(device)


def synthetic_conditioning(model, B, L, device, real: bool):
    """(channels, embedding): dummy (one impulse at sample 0, zero embedding) or the SURVEY 8d seeded "real" conditioning."""
    import torch

    track = torch.zeros(B, 1, L, device=device)
    if real:
        g = torch.Generator().manual_seed(3000)
        for b in range(B):
            k = int(torch.randint(1, 9, (1,), generator=g))
            track[b, 0, torch.randint(0, L, (k,), generator=g).to(device)] = 1.0
        emb = torch.nn.functional.normalize(torch.randn(B, 1, 512, generator=torch.Generator().manual_seed(2000)), dim=-1).to(device)
    else:
        track[:, 0, 0] = 1.0
        emb = torch.zeros(B, 1, 512, device=device)
    with torch.no_grad():      # the inference engine (with autograd recording Encoder1d runs the differentiable fp32 composition)
        _, info = model.onsets_encoder(track, with_info=True)
    return info["xs"][2:-1], emb


REPEATS = 3   # every secondary leg times its loop this many times: boxes differ by +-2-3 %, two runs on one box by < 0.5 %


def spread(rates) -> dict:
    """min / median / max of the repeated timings of a leg (steps/s unless the leg says otherwise)."""
    v = sorted(rates)
    return dict(min=round(v[0], 3), median=round(v[len(v) // 2], 3), max=round(v[-1], 3), repeats=len(v))


_SIDE = {}


def clock_probe_start(device, microseconds: float) -> None:
    """One wave on a side stream compares the shader-cycle counter with the 100 MHz counter while the timed loop runs (sf_clock_probe_*)."""
    import torch

    from syncfusion_amd import _lib

    if "stream" not in _SIDE:
        _SIDE["stream"] = torch.cuda.Stream(device)
    _lib.check(_lib.load().sf_clock_probe_start(float(microseconds), int(_SIDE["stream"].cuda_stream)), "sf_clock_probe_start")


def clock_probe_read() -> float:
    import ctypes

    from syncfusion_amd import _lib

    mhz = ctypes.c_double()
    _lib.check(_lib.load().sf_clock_probe_read(ctypes.byref(mhz)), "sf_clock_probe_read")
    return round(mhz.value, 1)


def timed_sample(model, device, noise, channels, emb, scale, steps, warm=2, repeats=REPEATS, clock=False):
    """(median steps/s of `repeats` timed sample() calls, last output, spread dict).  clock: the LAST repetition runs with the in-kernel
    clock probe beside it (spread["shader_mhz_during_last"]); its rate is reported separately and not part of the spread."""
    import torch

    def run(n):
        return model.model.sample(x_noisy=noise, num_steps=n, channels=channels, embedding=emb, embedding_scale=scale)

    run(2)
    if warm:
        run(warm)
    rates = []
    for _ in range(repeats):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        out = run(steps)
        torch.cuda.synchronize(device)
        rates.append(steps / (time.perf_counter() - t0))
    sp = spread(rates)
    if clock:
        torch.cuda.synchronize(device)
        clock_probe_start(device, 0.8 * 1e6 * steps / sp["median"])     # watches the first 80 % of the loop
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize(device)
        sp["probed_run_steps_per_s"] = round(steps / (time.perf_counter() - t0), 3)
        sp["shader_mhz_during_probed_run"] = clock_probe_read()
    return sp["median"], out, sp


def extra_workloads(model, device, args, noise, channels, emb) -> dict:
    """Secondary numbers (driver-visible, same process).  Each leg is independent: a failure is recorded, not raised."""
    import torch

    from syncfusion_amd import workmodel

    net = model.model.net
    hp = dict(net.hparams)
    out: dict = {}

    stash: dict = {}    # (noise, channels, embedding, scale, steps, final sample) of the 16-bit legs, for their parity-grade twins

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 -- the headline line must survive a failing secondary leg
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.synchronize(device)

    def roof_ms(clips, evals, L, dtype):
        w = workmodel.unet_work(hp, L, clips, evals, ES[dtype])
        peak = {"fp32": PEAK_F32_TFLOPS, "fp32x": PEAK_X3_TFLOPS}.get(dtype, PEAK_BF16_TFLOPS) * 1e12
        return workmodel.step_roofline_ms(w, peak, PEAK_HBM_GBS * 1e9)

    import contextlib

    @contextlib.contextmanager
    def engine_dtype(dt):
        """the same torch module with its engine repacked in another arithmetic (identical fp32 master weights)"""
        prev = net.compute_dtype
        net.compute_dtype = dt
        try:
            yield
        finally:
            net.compute_dtype = prev
            net.engine()

    def rel_l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    def parity_twin(dt, nz, ch, e, scale, steps, lowp_out, L):
        """The workload of a 16-bit leg again on a parity-grade engine (`dt` = fp32 or fp32x): its rate, its roofline fraction against the MFMA
        peak of ITS arithmetic (fp32 MFMA for fp32; a third of the 16-bit peak for fp32x: three 16-bit products per product), and the distance of the 16-bit engine's final sample from it after the configuration's own step count."""
        with engine_dtype(dt):
            rate, o, sp = timed_sample(model, device, nz, ch, e, scale, steps, warm=2, repeats=REPEATS)
        ms = 1e3 / rate
        B_ = nz.shape[0]
        return dict(steps_per_s=round(rate, 2), steps_per_s_spread=sp, ms_per_step=round(ms, 3), dtype=dt, timed_steps=steps,
                    step_roofline_frac=round(roof_ms(B_, 1 if scale == 1.0 else 2, L, dt) / ms, 4),
                    step_roofline_peak_tflops={"fp32": PEAK_F32_TFLOPS, "fp32x": PEAK_X3_TFLOPS}[dt],
                    lowp_dtype=args.dtype, lowp_final_sample_rel_l2=float(f"{rel_l2(lowp_out, o):.3e}"), rel_l2_steps=steps,
                    gate_1e4_met_by_lowp=bool(rel_l2(lowp_out, o) < 1e-4)), o

    PARITY_DTYPES = [d for d in ("fp32", "fp32x") if d in __import__("syncfusion_amd")._lib.DTYPES]

    def with_twins(res, nz, ch, e, scale, steps, lowp_out, L):
        outs = {}
        for dt in PARITY_DTYPES:
            try:
                res[dt], outs[dt] = parity_twin(dt, nz, ch, e, scale, steps, lowp_out, L)
            except Exception as ex:  # noqa: BLE001
                res[dt] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        if "fp32" in outs and "fp32x" in outs:
            res["fp32x"]["final_sample_rel_l2_vs_fp32"] = float(f"{rel_l2(outs['fp32x'], outs['fp32']):.3e}")
            res["fp32x"]["speedup_vs_fp32"] = round(res["fp32x"]["steps_per_s"] / res["fp32"]["steps_per_s"], 3)
        return res

    def fp32_leg(pdt="fp32"):
        # the same torch module, engine repacked in fp32 / fp32x (identical weights): the paths gated at 1e-4 against the oracle
        steps = 40   # (10 timed steps gave 179-209 steps/s from run to run on boxes where 30 give 216 three times in a row)
        ref_steps = min(args.steps, 20)
        lo = model.model.sample(x_noisy=noise, num_steps=ref_steps, channels=channels, embedding=emb, embedding_scale=args.scale)
        with engine_dtype(pdt):
            rate, _, sp = timed_sample(model, device, noise, channels, emb, args.scale, steps, warm=2)
            hi = model.model.sample(x_noisy=noise, num_steps=ref_steps, channels=channels, embedding=emb, embedding_scale=args.scale)
        rel = rel_l2(lo, hi)
        ms = 1e3 / rate
        res = dict(steps_per_s=round(rate, 2), steps_per_s_spread=sp, ms_per_step=round(ms, 3), dtype=pdt, batch=noise.shape[0], timed_steps=steps,
                   step_roofline_frac=round(roof_ms(noise.shape[0], 1 if args.scale == 1.0 else 2, L0, "fp32") / ms, 4),
                   lowp_vs_fp32_final_sample_rel_l2=round(rel, 6), lowp_dtype=args.dtype, rel_l2_steps=ref_steps)
        if pdt == "fp32":
            stash["fp32_config1"] = hi
        elif "fp32_config1" in stash:
            res["final_sample_rel_l2_vs_fp32"] = float(f"{rel_l2(hi, stash.pop('fp32_config1')):.3e}")
        return res

    def config2_leg():
        B, scale, steps = 32, 2.0, BASELINE_STEPS      # the configuration's own 50 steps: the per-call conditioning is amortised as in a real run
        nz = torch.randn(B, 1, L0, generator=torch.Generator().manual_seed(1000)).to(device)
        ch, e = synthetic_conditioning(model, B, L0, device, real=True)
        rate, o, sp = timed_sample(model, device, nz, ch, e, scale, steps, warm=2)
        assert torch.isfinite(o).all()
        stash["config2"] = (nz, ch, e, scale, steps, o)
        ms = 1e3 / rate
        w = workmodel.unet_work(hp, L0, B, 2, ES[args.dtype])
        return dict(workload="BASELINE configs[2]: batch=32, guidance scale 2.0 (64 evaluations/step), CLAP-shaped embedding + onset conditioning",
                    steps_per_s=round(rate, 2), steps_per_s_spread=sp, ms_per_step=round(ms, 3), dtype=args.dtype, timed_steps=steps,
                    algorithmic_tflop_per_step=round(w["flops"] / 1e12, 3), tflops=round(w["flops"] / 1e12 / (ms * 1e-3), 1),
                    step_roofline_frac=round(roof_ms(B, 2, L0, args.dtype) / ms, 4))

    def config3_leg():
        # BASELINE configs[3]: 256 clips over 8 GPUs = 32 clips per GPU, no guidance: one GPU's share (the N-GPU run itself is the
        # driver's `--gpus N` weak-scaling sweep of configs[1])
        B, steps = 32, BASELINE_STEPS
        nz = torch.randn(B, 1, L0, generator=torch.Generator().manual_seed(1000)).to(device)
        ch, e = synthetic_conditioning(model, B, L0, device, real=True)
        rate, o, sp = timed_sample(model, device, nz, ch, e, 1.0, steps, warm=2)
        assert torch.isfinite(o).all()
        stash["config3"] = (nz, ch, e, 1.0, steps, o)
        ms = 1e3 / rate
        w = workmodel.unet_work(hp, L0, B, 1, ES[args.dtype])
        return dict(workload="BASELINE configs[3], one GPU's share: batch=32 (256 clips / 8 GPUs), no guidance", steps_per_s=round(rate, 2), steps_per_s_spread=sp,
                    ms_per_step=round(ms, 3), clip_steps_per_s=round(rate * B, 1), dtype=args.dtype, timed_steps=steps,
                    tflops=round(w["flops"] / 1e12 / (ms * 1e-3), 1), step_roofline_frac=round(roof_ms(B, 1, L0, args.dtype) / ms, 4))

    def reference_leg():
        B, L, scale, steps = 10, 262144, 2.0, 20     # exp/evaluate_gh_gen.yaml:8 (length), :21 (batch_size), :23 (embedding_scale)
        nz = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(1000)).to(device)
        ch, e = synthetic_conditioning(model, B, L, device, real=True)
        rate, o, sp = timed_sample(model, device, nz, ch, e, scale, steps, warm=2)
        assert torch.isfinite(o).all()
        ms = 1e3 / rate
        w = workmodel.unet_work(hp, L, B, 2, ES[args.dtype])
        return dict(workload="reference evaluation shape: batch=10, length=2**18, embedding_scale=2.0 (exp/evaluate_gh_gen.yaml:8,21-23)",
                    steps_per_s=round(rate, 2), steps_per_s_spread=sp, ms_per_step=round(ms, 3), dtype=args.dtype, timed_steps=steps,
                    tflops=round(w["flops"] / 1e12 / (ms * 1e-3), 1), step_roofline_frac=round(roof_ms(B, 2, L, args.dtype) / ms, 4))

    def onset_leg():
        from syncfusion_amd.onset_net import VideoOnsetNet

        N, iters = 32, 10
        frames = torch.randn(N, 3, 30, 112, 112, generator=torch.Generator().manual_seed(4000)).to(device)
        res = {}
        # bf16 (the headline's arithmetic) and fp16 (what BASELINE configs[4] names, and the type that keeps the onset track index-identical)
        for dt in (("bf16", "fp16") if args.dtype != "fp16" else ("fp16",)):
            torch.manual_seed(7)
            onset = VideoOnsetNet(False, dtype=dt).to(device).eval()
            for _ in range(3):   # weight packing on the first call; clocks / caches settle on this workload after the U-Net legs
                onset(frames)
            times = []
            for _ in range(REPEATS):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(iters):
                    y = onset(frames)
                torch.cuda.synchronize(device)
                times.append((time.perf_counter() - t0) / iters)
            assert y.shape == (N, 30) and torch.isfinite(y).all()
            tfs = [N * workmodel.ONSET_NET_GFLOP_PER_CLIP / 1e3 / t for t in times]
            sp = spread(tfs)
            res[dt] = dict(clips_per_s=round(N * sp["median"] * 1e3 / (N * workmodel.ONSET_NET_GFLOP_PER_CLIP), 1), tflops=round(sp["median"], 1),
                           tflops_spread=sp, mfma_frac=round(sp["median"] / PEAK_BF16_TFLOPS, 4), dtype=dt)
            del onset
        first = res["bf16"] if "bf16" in res else res["fp16"]
        return dict(workload="VideoOnsetNet (R(2+1)D-18) forward, N=32 clips of (3,30,112,112)", **first, by_dtype=res)

    def e2e_leg(udt="fp16"):
        # BASELINE configs[4] on one GPU's share (32 clips): 2 s x 15 fps RGB frames -> onset net -> logits-to-track glue ->
        # Encoder1d -> 100-step guided diffusion -> cut_prefix / crop, everything in fp16, no host round trip in between
        from syncfusion_amd.generation import generate_batch
        from syncfusion_amd.onset_glue import onsets_to_track
        from syncfusion_amd.onset_net import VideoOnsetNet

        B, steps, scale = 32, 100, 2.0
        torch.manual_seed(7)
        onset = VideoOnsetNet(False, dtype="fp16").to(device).eval()
        frames = torch.randn(B, 3, 30, 112, 112, generator=torch.Generator().manual_seed(4000)).to(device)
        z = (torch.randn(B, 1, L0, generator=torch.Generator().manual_seed(1)) * 0.1).to(device)
        prev = net.compute_dtype
        net.compute_dtype = udt
        try:
            def once(n_steps):
                logits = onset(frames)
                logits[:, 3] = 1.0   # random-init logits sit near 0.1 (no onsets): force one so cut_prefix has a first onset
                track = onsets_to_track(logits, L0, frame_rate=15.0, sample_rate=22528.0)
                return generate_batch(model, track, z, num_steps=n_steps, length=L0, embedding_scale=scale, cut_prefix=True, cut_length=44100)

            once(2)
            secs = []
            for _ in range(REPEATS):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                gen = once(steps)
                torch.cuda.synchronize(device)
                secs.append(time.perf_counter() - t0)
            dt = sorted(secs)[len(secs) // 2]
        finally:
            net.compute_dtype = prev
            net.engine()
        assert gen.shape == (B, 1, 44100) and torch.isfinite(gen).all()
        res = dict(workload="BASELINE configs[4], one GPU's share: 32 clips of 30x112x112 RGB frames -> VideoOnsetNet (fp16) -> onset track -> Encoder1d -> "
                            f"100-step diffusion (scale 2.0, {udt}) -> cut/crop", clips_per_s=round(B / dt, 2), seconds_per_batch=round(dt, 3),
                   seconds_per_batch_spread=spread(secs),
                   denoise_steps_per_s=round(steps / dt, 2), dtype=udt)
        if udt == "fp16":
            stash["e2e"] = gen
        elif "e2e" in stash:
            r = rel_l2(stash["e2e"], gen)
            res.update(fp16_final_audio_rel_l2=float(f"{r:.3e}"), gate_1e4_met_by_fp16=bool(r < 1e-4))
            stash["e2e_" + udt] = gen
            if udt == "fp32x" and "e2e_fp32" in stash:
                res["final_audio_rel_l2_vs_fp32"] = float(f"{rel_l2(gen, stash['e2e_fp32']):.3e}")
        return res

    def parity_legs():
        """The parity-grade engines (fp32: v_mfma_f32_32x32x2_f32; fp32x: fp32-accurate products from split 16-bit operands) at EVERY BASELINE
        configuration the 16-bit legs above ran: steps/s, roofline fraction against the fp32 MFMA peak, and the 16-bit engine's final-sample
        rel-L2 against them after the configuration's own step count (north_star's gate is 1e-4)."""
        res = {}
        for name, key, L in (("config2_b32_cfg", "config2", L0), ("config3_share_b32", "config3", L0)):
            if key in stash:
                nz, ch, e, scale, steps, o = stash.pop(key)
                res[name] = with_twins({}, nz, ch, e, scale, steps, o, L)
                del nz, ch, e, o
        for dt in PARITY_DTYPES:
            try:
                res.setdefault("e2e_config4", {})[dt] = e2e_leg(dt)
            except Exception as ex:  # noqa: BLE001
                res.setdefault("e2e_config4", {})[dt] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        stash.clear()
        torch.cuda.empty_cache()
        return res

    def train_leg():
        # the reference's training configuration (exp/train_diffusion_gh.yaml:8,38,87): fp32, batch 4 per device, clips of 2^18
        # samples; Model.training_step -> loss.backward() (HIP forward + backward kernels) -> AdamW over U-Net + onset encoder
        B, L, iters = 4, 262144, 3 * REPEATS     # REPEATS timed groups of three steps after one untimed step
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 1, L, generator=g).to(device)
        y = (torch.rand(B, 1, L, generator=g) < 0.0005).float().to(device)
        def graph_part():
            # forward + backward captured once in a HIP graph (syncfusion_amd.training.GraphedTrainStep): the eager step is host-bound once
            # the GEMMs run on the split operands.  Runs FIRST: capture needs a process whose autograd nodes were not yet created on the
            # default stream by an eager backward.
            import gc

            from syncfusion_amd.training import GraphedTrainStep

            saved2 = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("clap.")}
            opt2 = model.configure_optimizers()
            gs = None
            try:
                with torch.enable_grad():
                    gs = GraphedTrainStep(model, (x, y, x, None, None))
                    glosses, marks = [], []
                    for it in range(iters + 1):
                        if it >= 1 and (it - 1) % 3 == 0:
                            torch.cuda.synchronize(device)
                            marks.append(time.perf_counter())
                        glosses.append(gs.step().detach().clone())
                        opt2.step()
                    torch.cuda.synchronize(device)
                    marks.append(time.perf_counter())
                gms = [1e3 * (b_ - a_) / 3 for a_, b_ in zip(marks[:-1], marks[1:])]
                glosses = [round(float(v), 5) for v in glosses]
                assert all(math.isfinite(v) for v in glosses) and glosses[-1] < glosses[0], glosses
                return dict(ms_per_step=round(sorted(gms)[len(gms) // 2], 2), ms_per_step_spread=spread(gms), losses=glosses,
                            note="forward + backward replayed from one HIP graph (training.GraphedTrainStep), AdamW eager")
            finally:
                del opt2, gs
                model.zero_grad(set_to_none=True)
                model.load_state_dict(saved2, strict=False)
                gc.collect()
                torch.cuda.synchronize(device)
                torch.cuda.empty_cache()

        try:
            graph_res = graph_part()
        except Exception as e:  # noqa: BLE001
            graph_res = {"error": f"{type(e).__name__}: {e}"[:300]}
        saved = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("clap.")}   # the step moves the weights
        opt = model.configure_optimizers()
        losses = []
        try:
            with torch.enable_grad():
                marks = []
                for it in range(iters + 1):
                    if it >= 1 and (it - 1) % 3 == 0:
                        torch.cuda.synchronize(device)
                        marks.append(time.perf_counter())
                    loss = model.training_step((x, y, x, None, None), it)
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    opt.step()
                    losses.append(loss.detach())   # read back after the timed region: a host read per step would serialise the
                                                   # Python-issued forward of step i+1 behind the backward of step i
                torch.cuda.synchronize(device)
                marks.append(time.perf_counter())
                group_ms = [1e3 * (b_ - a_) / 3 for a_, b_ in zip(marks[:-1], marks[1:])]
                dt = sorted(group_ms)[len(group_ms) // 2] / 1e3
                losses = [round(float(v), 5) for v in losses]
        finally:
            del opt
            model.zero_grad(set_to_none=True)
            model.load_state_dict(saved, strict=False)
            torch.cuda.empty_cache()
        assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
        from syncfusion_amd import autograd as sfa

        res = dict(workload="training step (exp/train_diffusion_gh.yaml): fp32 tensors, batch 4 x 2**18 samples, v-objective loss -> backward -> AdamW; "
                            f"GEMM arithmetic {sfa.GEMM_DTYPE} (fp32x = products from split 16-bit operands, gradient tests at unchanged tolerances)",
                   dtype=sfa.GEMM_DTYPE, timed_steps=iters,
                   eager=dict(ms_per_step=round(1e3 * dt, 2), ms_per_step_spread=spread(group_ms), losses=losses,
                              note="Python issues ~4100 launches per step: host-bound on boxes with slow host cores"),
                   graph_replay=graph_res)
        # the step as a user runs it fastest: both modes are product API (Model.training_step + backward, or training.GraphedTrainStep)
        g_ms = graph_res.get("ms_per_step")
        if g_ms is not None and g_ms < 1e3 * dt:
            res.update(ms_per_step=g_ms, ms_per_step_spread=graph_res["ms_per_step_spread"], clips_per_s=round(B / (g_ms / 1e3), 2),
                       mode="graph replay (training.GraphedTrainStep: forward + backward from one HIP graph, AdamW eager)")
        else:
            res.update(ms_per_step=round(1e3 * dt, 2), ms_per_step_spread=spread(group_ms), clips_per_s=round(B / dt, 2), mode="eager")
        return res

    def transpose_up_legs():
        # The OTHER candidate network (SURVEY 8f-1 cannot be settled offline): upsample_mode="transpose", a-unet's `Upsample` =
        # ConvTranspose1d(kernel = stride = factor), north_star's "transposed-conv blocks".  Same seed, same inputs, same timing
        # protocol as the headline (configs[1]) and as config2_b32_cfg, so both candidate networks are driver-timed.
        mt = build_model(args.dtype, device, upsample_mode="transpose")
        mt.model.sampler.use_graph = not args.no_graph
        hpt = dict(mt.model.net.hparams)
        res = {}
        try:
            B1 = noise.shape[0]
            ch1, e1 = synthetic_conditioning(mt, B1, L0, device, real=False)
            rate, o, sp = timed_sample(mt, device, noise, ch1, e1, 1.0, BASELINE_STEPS, warm=5)
            assert torch.isfinite(o).all()
            w = workmodel.unet_work(hpt, L0, B1, 1, ES[args.dtype], upsample_mode="transpose")
            peak = {"fp32": PEAK_F32_TFLOPS, "fp32x": PEAK_X3_TFLOPS}.get(args.dtype, PEAK_BF16_TFLOPS) * 1e12
            ms = 1e3 / rate
            res["config1_transpose_up"] = dict(
                workload=f"BASELINE configs[1] on the transposed-up network: batch={B1}, {BASELINE_STEPS} steps, scale 1.0, dummy cond",
                steps_per_s=round(rate, 2), steps_per_s_spread=sp, ms_per_step=round(ms, 3), dtype=args.dtype, timed_steps=BASELINE_STEPS,
                params_M=round(sum(p.numel() for p in mt.model.net.parameters()) / 1e6, 2),
                step_roofline_frac=round(workmodel.step_roofline_ms(w, peak, PEAK_HBM_GBS * 1e9) / ms, 4))
            B2, scale = 32, 2.0
            nz = torch.randn(B2, 1, L0, generator=torch.Generator().manual_seed(1000)).to(device)
            ch2, e2 = synthetic_conditioning(mt, B2, L0, device, real=True)
            rate, o, sp = timed_sample(mt, device, nz, ch2, e2, scale, BASELINE_STEPS, warm=2)
            assert torch.isfinite(o).all()
            w = workmodel.unet_work(hpt, L0, B2, 2, ES[args.dtype], upsample_mode="transpose")
            ms = 1e3 / rate
            res["config2_transpose_up"] = dict(
                workload="BASELINE configs[2] on the transposed-up network: batch=32, guidance scale 2.0, real conditioning",
                steps_per_s=round(rate, 2), steps_per_s_spread=sp, ms_per_step=round(ms, 3), dtype=args.dtype, timed_steps=BASELINE_STEPS,
                tflops=round(w["flops"] / 1e12 / (ms * 1e-3), 1),
                step_roofline_frac=round(workmodel.step_roofline_ms(w, peak, PEAK_HBM_GBS * 1e9) / ms, 4))
        finally:
            del mt
            torch.cuda.empty_cache()
        return res

    def config1_repeat_leg():
        # the headline's own loop (same inputs, same --steps) three more times: what a 2-6 % difference between rounds has to be read against
        rate, _, sp = timed_sample(model, device, noise, channels, emb, args.scale, args.steps, warm=0, clock=True)
        return dict(workload=f"the headline loop again AFTER the roofline / cpu_baseline legs (the engine was rebuilt once in between): batch={noise.shape[0]}, "
                             f"{args.steps} steps, scale {args.scale}", steps_per_s=round(rate, 2),
                    steps_per_s_spread=sp, dtype=args.dtype, timed_steps=args.steps)

    leg("config1_repeat", config1_repeat_leg)
    if args.dtype != "fp32":
        leg("fp32_config1", fp32_leg)
        if "fp32x" in PARITY_DTYPES:
            leg("fp32x_config1", lambda: fp32_leg("fp32x"))
    leg("config2_b32_cfg", config2_leg)
    try:
        out.update(transpose_up_legs())
    except Exception as e:  # noqa: BLE001
        out["config1_transpose_up"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.synchronize(device)
    leg("config3_share_b32", config3_leg)
    leg("reference_eval_shape", reference_leg)
    leg("onset_net_n32", onset_leg)
    leg("e2e_config4_fp16", e2e_leg)
    if args.dtype != "fp32":
        leg("parity_engines", parity_legs)
    leg("train_step_fp32", train_leg)      # last: it updates (and then restores) the weights
    return out


def main() -> int:
    args = parse_args()
    have_env = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_env:
        return spawn_ranks(args)            # before `import torch` touches the GPU: the parent stays a plain launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if have_env and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}", file=sys.stderr)
        return 2

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        if args.share_gpu:
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from syncfusion_amd import dist as sfd
    from syncfusion_amd import workmodel

    sfd.FORCE_COLLECTIVES = bool(args.force_dist)

    B = args.batch
    model = build_model(args.dtype, device)
    model.model.sampler.use_graph = not args.no_graph
    bcast_bytes = sfd.broadcast_module(model, src=0)          # RCCL broadcast of the weights, once, outside the timed loop
    net = model.model.net

    # synthetic inputs, identical bits on every run: per-rank noise seed 1000 + rank
    noise = torch.randn(B, 1, L0, generator=torch.Generator().manual_seed(sfd.rank_seed(1000, rank))).to(device)
    channels, emb = synthetic_conditioning(model, B, L0, device, real=False)   # "dummy cond" of configs[1]

    def run(steps):
        return model.model.sample(x_noisy=noise, num_steps=steps, channels=channels, embedding=emb, embedding_scale=args.scale)

    def fence():
        torch.cuda.synchronize(device)
        if dist_on:
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
        t = torch.tensor([elapsed], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    assert torch.isfinite(out).all()
    gathered = sfd.gather_clips(out, B * world, dst=0)         # RCCL gather of the finished clips, once, after the loop
    # The same loop three more times RIGHT AWAY (same engine object, same graphs, nothing else has run): what the headline has to be read
    # against.  `config1_repeat` in `extra` does the same after the roofline / cpu_baseline legs.
    repeat_early = None
    if rank == 0 and world == 1 and not args.no_extra and not args.force_dist:
        try:
            r_, _, sp_ = timed_sample(model, device, noise, channels, emb, args.scale, args.steps, warm=0, clock=True)
            repeat_early = dict(steps_per_s=round(r_, 2), steps_per_s_spread=sp_, timed_steps=args.steps)
        except Exception as e:  # noqa: BLE001
            repeat_early = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return 0

    # ---------------- roofline: per-kernel HIP-event timing of one instrumented evaluation ----------------
    evals = 2 if args.scale != 1.0 else 1
    sigma = torch.full((B,), 0.5, device=device)
    net.engine().profile_forward(noise, sigma, channels, emb, args.scale)          # warm
    recs5 = net.engine().profile_forward(noise, sigma, channels, emb, args.scale, with_depth=True)
    recs = [r[:4] for r in recs5]
    if args.dump_launches:
        with open(args.dump_launches, "w") as f:
            for i, (label, ms, fl, by) in enumerate(recs):
                f.write(f"{i}\t{label}\t{ms * 1e3:.2f}us\t{fl / 1e6:.2f}MFLOP\t{by / 1e6:.3f}MB\n")
    # HIP events bracket launch latency as well as execution: the engine times an empty kernel the same way
    # ("calib_empty"); its duration minus ~1.5 us of real execution is the fixed cost subtracted from every launch
    calib = sorted(ms for label, ms, _, _ in recs if label == "calib_empty")
    # Calibrated against rocprofv3 --kernel-trace of the same command (profiles/*_kernel_stats.csv): about half of the
    # empty-kernel event time overlaps with a real kernel's own launch ramp, so half of it is subtracted.
    overhead_ms = 0.5 * max(0.0, (calib[len(calib) // 2] if calib else 0.0) - 1.5e-3)
    recs = [(label, max(ms - overhead_ms, 1e-4), fl, by) for label, ms, fl, by in recs if label != "calib_empty"]
    depth_ms = defaultdict(float)      # device time of the instrumented evaluation by U-Net depth (-1: per-step features)
    depth_n = defaultdict(int)
    for label, ms, _, _, dep in recs5:
        if label != "calib_empty":
            depth_ms[dep] += max(ms - overhead_ms, 1e-4)
            depth_n[dep] += 1
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
    peak = {"fp32": PEAK_F32_TFLOPS, "fp32x": PEAK_X3_TFLOPS}.get(args.dtype, PEAK_BF16_TFLOPS)
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
    if args.batch == BATCH and args.dtype == "bf16" and args.scale == 1.0:
        import glob

        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[::-1]:
            try:
                with open(path) as f:
                    kern = json.load(f).get("kernels", {})
            except (OSError, ValueError):
                continue
            ent = kern.get(dom.split("<")[0])
            if ent:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
                break
    roof.update(traffic=traffic, traffic_source=traffic_src, algorithmic_bytes_per_launch=round(d_by / d_n), kernel=dom, launches_per_eval=d_n, avg_launch_us=round(d_ms / d_n * 1e3, 3),
                share_of_eval_time=round(d_ms / total_ms, 4),
                per_kernel={k: dict(ms=round(v[0], 4), launches=v[3], tflops=round(v[1] / max(v[0], 1e-9) / 1e9, 3),
                                    gbs=round(v[2] / max(v[0], 1e-9) / 1e6, 1)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])},
                eval_device_ms=round(total_ms, 4), launches_per_eval_total=len(recs),
                event_overhead_us_subtracted=round(overhead_ms * 1e3, 3))
    # whole-step roofline, closed form: t_roofline = sum_depth max(F_d / P_mfma, Q_d / BW_hbm) with every weight
    # counted once per step -- independent of the engine's clip-parallel branches and of its launch structure
    work = workmodel.unet_work(dict(net.hparams), L0, B, evals, ES[args.dtype])
    t_roof = workmodel.step_roofline_ms(work, peak * 1e12, PEAK_HBM_GBS * 1e9)
    ms_per_step = elapsed / args.steps * 1e3
    roof["step_algorithmic_tflop"] = round(work["flops"] / 1e12, 4)
    roof["step_algorithmic_gb"] = round(work["bytes"] / 1e9, 4)
    roof["step_roofline_ms"] = round(t_roof, 4)
    roof["step_roofline_frac"] = round(t_roof / ms_per_step, 5)
    roof["step_hbm_gbs_algorithmic"] = round(work["bytes"] / 1e9 / (ms_per_step * 1e-3), 1)
    # Per-depth-group roofline: depths 0-3 against the HBM peak, depths 4-7 against the dense MFMA peak.  Work = the
    # closed-form algorithmic bytes / FLOPs of those depths for one step (weights once); time = the group's share of the TIMED step
    # (ms_per_step apportioned by the device time of the group's launches in the instrumented evaluation, where the clip-parallel
    # branches run one after another); `frac_serial` prices the same work against the un-overlapped device time itself.
    nd = len(work["flops_by_depth"])
    split = min(4, nd)
    dev_total = sum(depth_ms.values())
    groups = {}
    for name, ds, bound in ((f"d0-{split - 1}", range(0, split), "hbm"), (f"d{split}-{nd - 1}", range(split, nd), "mfma")):
        if len(ds) == 0:
            continue
        g_ms = sum(depth_ms.get(d, 0.0) for d in ds)
        share_ms = ms_per_step * g_ms / max(dev_total, 1e-9)
        fl = sum(work["flops_by_depth"][d] for d in ds)
        by = sum(work["bytes_by_depth"][d] for d in ds)
        if bound == "hbm":
            a_step, a_ser, pk, unit = by / 1e9 / (share_ms * 1e-3), by / 1e9 / (g_ms * 1e-3), PEAK_HBM_GBS, "GB/s"
        else:
            a_step, a_ser, pk, unit = fl / 1e12 / (share_ms * 1e-3), fl / 1e12 / (g_ms * 1e-3), peak, "TFLOP/s"
        groups[name] = dict(bound=bound, achieved=round(a_step, 3), peak=pk, unit=unit, frac=round(a_step / pk, 5),
                            frac_serial=round(a_ser / pk, 5), step_share_ms=round(share_ms, 4), device_ms_serial=round(g_ms, 4),
                            launches=sum(depth_n.get(d, 0) for d in ds), algorithmic_gb=round(by / 1e9, 4),
                            algorithmic_tflop=round(fl / 1e12, 4))
    roof["depth_groups"] = groups

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
        # BASELINE configs[0] IN FULL: 1 clip, 10 sampler steps, guidance scale 2.0, L0 = 45056, the call
        # shape of main/module_diffusion.py:200-206 -- the oracle's whole sample() on the host cores, and the SAME call on the HIP
        # fp32 engine next to it (identical noise), with the distance between the two final samples.
        if not args.no_config0:
            from oracle import sampler_ref

            steps0, scale0 = 10, 2.0
            nz0 = torch.randn(1, 1, L0, generator=torch.Generator().manual_seed(1000))
            ch0 = [c[:1].cpu() for c in channels]
            e0 = torch.nn.functional.normalize(torch.randn(1, 1, 512, generator=torch.Generator().manual_seed(2000)), dim=-1)
            with torch.no_grad():
                t1 = time.perf_counter()
                ref0 = sampler_ref.vsample(lambda x, s_: unet_ref.unet_forward(P, cfg, x, s_, embedding=e0, channels=ch0, embedding_scale=scale0), nz0, steps0)
                cpu_s = time.perf_counter() - t1
            prev = net.compute_dtype
            net.compute_dtype = "fp32"
            try:
                gch0 = [c.to(device) for c in ch0]
                kw0 = dict(x_noisy=nz0.to(device), num_steps=steps0, channels=gch0, embedding=e0.to(device), embedding_scale=scale0)
                model.model.sample(**kw0)
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                got0 = model.model.sample(**kw0)
                torch.cuda.synchronize(device)
                gpu_s = time.perf_counter() - t1
            finally:
                net.compute_dtype = prev
                net.engine()
            rel0 = float((got0.cpu().double() - ref0.double()).norm() / ref0.double().norm())
            cpu["config0_full"] = dict(workload="BASELINE configs[0]: 1 clip, 10 steps, scale 2.0 (20 U-Net evaluations), L0 45056, fp32",
                                       seconds=round(cpu_s, 3), steps_per_s=round(steps0 / cpu_s, 4), cores=torch.get_num_threads(), kind="port",
                                       hip_fp32_engine_seconds=round(gpu_s, 4), hip_fp32_steps_per_s=round(steps0 / gpu_s, 2),
                                       rel_l2_hip_fp32_vs_cpu=float(f"{rel0:.3e}"))

    extra = None
    if not args.no_extra and world == 1 and not args.force_dist:
        extra = extra_workloads(model, device, args, noise, channels, emb)
        extra["config1_repeat_early"] = repeat_early

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
        "config": {"workload": f"BASELINE configs[1]: batch={B}/GPU, {BASELINE_STEPS}-step DDIM (v-sampler) denoising of 2 s @ 22.05 kHz clips "
                               f"(L0={L0} = 44*1024, cropped to 44100), {args.dtype}, embedding_scale={args.scale} ({evals} U-Net eval/step), "
                               f"dummy cond; this run timed {args.steps} consecutive sampler steps (the cost of a step does not depend on the "
                               "schedule length)",
                   "batch_per_gpu": B, "L0": L0, "evals_per_step": evals, "clip_steps_per_s": round(world * B * args.steps / elapsed, 2),
                   "params_M": round(sum(p.numel() for p in net.parameters()) / 1e6, 2), "hip_graph": not args.no_graph,
                   "weights_broadcast_bytes": bcast_bytes, "gathered_clips": None if gathered is None else int(gathered.shape[0]),
                   **({"dist_backend": dist.get_backend(), "ranks_share_a_gpu": bool(args.share_gpu)} if dist_on else {})},
        "roofline": roof,
        "cpu_baseline": cpu,
        "extra": extra,
    }
    print(json.dumps(line))
    sys.stdout.flush()
    if dist_on:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
