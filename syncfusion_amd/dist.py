"""Data-parallel sharding of independent clips across the GPUs of one node.

Clips never interact during sampling or in the onset net, so there is NO data-path collective:
each rank owns a contiguous slice of the batch.  The only collectives (RCCL over xGMI through
``torch.distributed`` backend ``nccl``; ``gloo`` in the CPU tests) are
  * ``broadcast_module``  -- rank 0's weights to every rank, once, as ONE flat buffer (one large transfer
    per peer instead of hundreds of small ones);
  * ``gather_clips``      -- each rank's finished clips to rank 0, once, after the loop.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``total`` clips owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank noise seed: an N-GPU run equals the concatenation of N single-GPU runs."""
    return base_seed + rank


# test hook (bench.py --force-dist): run the collectives even when the group has a single rank
FORCE_COLLECTIVES = False


def _single() -> bool:
    return (not (dist.is_available() and dist.is_initialized())) or (dist.get_world_size() == 1 and not FORCE_COLLECTIVES)


@torch.no_grad()
def broadcast_module(module: torch.nn.Module, src: int = 0) -> int:
    """Flat-buffer broadcast of every parameter and buffer from ``src``.  Returns bytes moved per rank."""
    if _single():
        return 0
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    if not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    if flat.is_cuda and dist.get_backend() != "nccl":      # gloo (CPU tests with device tensors): stage through the host
        host = flat.cpu()
        dist.broadcast(host, src=src)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off: off + n].reshape(t.shape).to(t.dtype))
        off += n
    return flat.numel() * 4


@torch.no_grad()
def gather_clips(local: torch.Tensor, total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank ``(b_r, ...)`` clip tensors into ``(total, ...)`` on ``dst`` (None elsewhere)."""
    if _single():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = local.device
    if local.is_cuda and dist.get_backend() != "nccl":     # gloo: gather on the host, return on the caller's device
        out = gather_clips(local.cpu(), total, dst)
        return None if out is None else out.to(dev)
    sizes = [shard_range(total, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxb,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs: Optional[List[torch.Tensor]] = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    if dist.get_backend() == "nccl":
        # RCCL has no native gather for uneven shapes; all_gather of the padded slice is one collective
        allb = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(allb, pad)
        bufs = allb if rank == dst else None
    else:
        dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
