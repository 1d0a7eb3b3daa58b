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


def rank_reports(dist, device_ordinal: int, status: np.ndarray, device=None) -> list:
    """Per rank, on every rank: the HIP device ordinal its handle is bound to and the status counts of its shard
    (solved, solved-inaccurate, infeasible, anything else)."""
    mine = [int(device_ordinal), int(np.sum(status == 1)), int(np.sum(status == 2)), int(np.sum(status == -3)),
            int(np.sum((status != 1) & (status != 2) & (status != -3)))]
    if dist is None:
        rows = [mine]
    else:
        import torch
        t = torch.tensor(mine, dtype=torch.int64, device=device or "cpu")
        parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        rows = [[int(v) for v in p.cpu().tolist()] for p in parts]
    return [{"rank": r, "device": row[0], "solved": row[1], "solved_inaccurate": row[2], "infeasible": row[3], "other": row[4]}
            for r, row in enumerate(rows)]


def gather_check(dist, rank: int, u0: np.ndarray, status: np.ndarray, total: int, solve_whole, device=None):
    """One result buffer: the shards' (u0, status) all-gathered on every rank, and - on rank 0 - checked against ONE
    process solving the whole batch (`solve_whole()` -> object with .u0, .status).  None without a process group of
    more than one rank; the dict of bench.py's `gather_check` entry on rank 0; {} on the other ranks."""
    if dist is None or dist.get_world_size() < 2:
        return None
    u_all, s_all = gather_controls(dist, u0, status, total, device=device)
    if rank != 0:
        return {}
    one = solve_whole()
    okm = (one.status == 1) & (s_all == 1)
    return {"instances": int(total), "status_equal": bool(np.array_equal(one.status, s_all)),
            "max_abs_u_diff": float(np.max(np.abs(one.u0[okm] - u_all[okm]))) if okm.any() else None,
            "note": "all-gathered (u0, status) of the %d shards vs one process solving the whole batch" % dist.get_world_size()}
