"""world_size-2 gloo test of the N>1 path: contiguous clip sharding, flat weight broadcast, output gather.

The data path has no collective (clips are independent); this covers the only two collectives the
multi-GPU run uses, on CPU tensors with the gloo backend.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank: int, world: int, port: int, total: int, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syncfusion_amd.dist import broadcast_module, gather_clips, rank_seed, shard_range

        torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
        lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5))
        moved = broadcast_module(lin, src=0)
        flat = torch.cat([p.detach().reshape(-1) for p in lin.parameters()])
        lo, hi = shard_range(total, rank, world)
        # every rank produces "clips" that depend only on the global clip index and its own seeded noise
        g = torch.Generator().manual_seed(rank_seed(1000, rank))
        local = torch.stack([torch.full((1, 6), float(i)) for i in range(lo, hi)]) + 0.0 * torch.randn(hi - lo, 1, 6, generator=g)
        out = gather_clips(local, total, dst=0)
        q.put((rank, moved, flat.tolist(), None if out is None else out[:, 0, 0].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, world, port = 5, 2, 29611
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, moved0, w0, out0), (_, moved1, w1, out1) = res
    assert w0 == w1, "weights differ after broadcast"
    assert moved0 == moved1 > 0
    assert out1 is None
    assert out0 == [0.0, 1.0, 2.0, 3.0, 4.0], "gathered clips are not the rank-order concatenation"


def test_shard_range_partitions_every_total_exactly():
    """Contiguous shards cover [0, total) once, in rank order, sizes differing by at most one -- for every world size of one node and
    totals around the configuration's 256 clips (uneven ones included)."""
    from syncfusion_amd.dist import shard_range

    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 7, 8, 9, 250, 255, 256, 257):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes


@pytest.mark.timeout(600)
def test_broadcast_and_gather_world8_uneven_total():
    """The configuration that will eventually run (BASELINE configs[3]: 8 ranks of one node) with a clip count that does NOT divide:
    250 clips over 8 ranks = two ranks of 32 and six of 31.  Flat weight broadcast from rank 0, every rank produces its own shard, the
    padded gather returns the rank-order concatenation on rank 0 only (main/generation.py:16,44 is single-device: the split is ours)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, world, port = 250, 8, 29617
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    weights = [r[2] for r in res]
    assert all(w == weights[0] for w in weights), "weights differ after broadcast"
    assert len({r[1] for r in res}) == 1 and res[0][1] > 0
    assert all(r[3] is None for r in res[1:])
    assert res[0][3] == [float(i) for i in range(total)], "gathered clips are not the rank-order concatenation"


def _grad_worker(rank: int, world: int, port: int, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from syncfusion_amd.training import allreduce_gradients

        torch.manual_seed(7)                                # same weights everywhere, DIFFERENT data per rank
        net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 3))
        net[2].bias.requires_grad_(False)                   # frozen tensors are skipped
        net.add_module("unused", torch.nn.Linear(2, 2))     # trainable, never in the graph: no gradient on any rank
        net.add_module("rank1_only", torch.nn.Linear(3, 3, bias=False))   # in the graph of rank 1 only
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(50 + rank))
        y = net[2](net[1](net[0](x)))
        if rank == 1:
            y = y + net.rank1_only(y.detach())
        y.square().mean().backward()
        calls = allreduce_gradients(net, bucket_bytes=200)  # tiny buckets: several collectives, one oversize tensor alone in its own
        none = sorted(n for n, p in net.named_parameters() if p.requires_grad and p.grad is None)
        q.put((rank, calls, torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None]).tolist(), none))
    finally:
        dist.destroy_process_group()


@pytest.mark.autograd
@pytest.mark.timeout(300)
def test_gradient_allreduce_world2_equals_the_full_batch_gradient():
    """Data-parallel training exchange (syncfusion_amd/training.py): bucketed mean of .grad over 2 gloo ranks == the gradient of
    the mean loss over both ranks' batches computed in one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29613
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, calls0, g0, none0), (_, calls1, g1, none1) = res
    assert g0 == g1 and calls0 == calls1 >= 2
    # a parameter without a gradient on EVERY rank keeps grad = None (as a single-rank run leaves it: the optimizer skips it);
    # one with a gradient on some rank gets the average on all of them
    assert none0 == none1 == ["unused.bias", "unused.weight"]
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 3))
    net[2].bias.requires_grad_(False)
    net.add_module("unused", torch.nn.Linear(2, 2))
    net.add_module("rank1_only", torch.nn.Linear(3, 3, bias=False))
    xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]
    ys = [net[2](net[1](net[0](x))) for x in xs]
    ys[1] = ys[1] + net.rank1_only(ys[1].detach())
    (sum(y.square().mean() for y in ys) / world).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])
    assert torch.allclose(torch.tensor(g0), ref, rtol=1e-5, atol=1e-7)
    from syncfusion_amd.training import allreduce_gradients

    assert allreduce_gradients(net) == 0                    # no process group: a no-op


_FAKE_RANK = r'''
import json, os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
print("noise on stdout from rank", dist.get_rank())
if dist.get_rank() == 0:
    sys.stdout.write("[partial banner without a newline] ")      # what an unbuffered C++ writer on the shared pipe can do to the line
    print(json.dumps({"metric": "fake", "n_gpus": dist.get_world_size(), "sum": float(t), "argv": sys.argv[1:]}))
dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
def test_bench_gpus_flag_spawns_fresh_ranks(tmp_path, capfd):
    """`bench.py --gpus N` from a plain shell (no WORLD_SIZE) starts N ranks under torch.distributed.run and relays rank 0's
    JSON line (the launcher itself, with a CPU stand-in for the rank body); a WORLD_SIZE that disagrees with --gpus is refused."""
    import json
    import subprocess

    import bench

    script = tmp_path / "fake_rank.py"
    script.write_text(_FAKE_RANK)
    import socket

    with socket.socket() as sock:                     # a port that is free right now (a fixed one can sit in TIME_WAIT from an earlier run)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    args = bench.parse_args(["--gpus", "2", "--master-port", str(port)])
    env_before = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        rc = bench.spawn_ranks(args, script=str(script), argv=["--gpus", "2", "--steps", "3"])
    finally:
        for k, v in env_before.items():
            if v is not None:
                os.environ[k] = v
    out, _ = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out                       # stdout carries exactly ONE JSON line; everything else went to stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--gpus", "2", "--steps", "3"]
    # a rank environment that disagrees with --gpus is an error, not a silent single-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 2 and "disagrees with WORLD_SIZE" in r.stderr
