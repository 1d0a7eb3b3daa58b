"""Process-group helpers of bench.py and the multi-rank tests (torch.distributed: backend "nccl" is RCCL on
ROCm, the CPU tests run the same code over "gloo").  Not part of the product package: the data path of the
sharded solve has no collective (multi-purpose-mpc_amd/sharding.py), these are the bookkeeping calls around it -
a MAX of the elapsed time, a gather of the per-rank times, and the gather of the 16-byte-per-instance controls
into one result buffer."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-purpose-mpc_amd"))
from sharding import shard_bounds  # noqa: E402


def max_over_ranks(dist, value: float, device=None) -> float:
    """MAX-reduce a host scalar (elapsed seconds) over the process group; identity without one."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(dist, value: float, device=None) -> list:
    """Every rank's host scalar, in rank order, on every rank."""
    if dist is None:
        return [float(value)]
    import torch
    mine = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


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
