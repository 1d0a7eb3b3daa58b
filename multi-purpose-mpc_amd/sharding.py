"""Batch sharding across the GPUs of one node.

The QP instances are independent, so the batch is cut into contiguous shards, one process + one
handle + one stream per device, and nothing is exchanged on the data path.  The only collective
is the bookkeeping the caller asks for: a barrier, a MAX over ranks of the elapsed time, and - when
one result buffer is wanted - an all-gather of the 16-byte-per-instance control outputs.
Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, as-even-as-possible split of `total` instances: [lo, hi) of `rank`."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(arrays, world: int, rank: int):
    """Slice every per-instance array (leading dimension = batch) to this rank's shard."""
    lo, hi = shard_bounds(arrays[0].shape[0], world, rank)
    return [a[lo:hi] for a in arrays]


def max_over_ranks(dist, value: float, device=None) -> float:
    """MAX-reduce a host scalar (elapsed seconds) over the process group; identity without one."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_controls(dist, u0: np.ndarray, status: np.ndarray, total: int, device=None):
    """All-gather the per-instance controls (v, delta) and statuses of every shard into one
    [total, 2] / [total] pair on every rank (shards may differ in size by one)."""
    if dist is None:
        return u0, status
    import torch
    world = dist.get_world_size()
    width = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world))
    pad = np.zeros((width, 3))
    pad[:u0.shape[0], :2] = u0
    pad[:u0.shape[0], 2] = status
    mine = torch.from_numpy(pad).to(device or "cpu")
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out_u, out_s = np.zeros((total, 2)), np.zeros(total, np.int32)
    for r, part in enumerate(parts):
        lo, hi = shard_bounds(total, world, r)
        block = part.cpu().numpy()[:hi - lo]
        out_u[lo:hi] = block[:, :2]
        out_s[lo:hi] = block[:, 2].astype(np.int32)
    return out_u, out_s
