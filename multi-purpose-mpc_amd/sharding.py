"""Batch sharding across the GPUs of one node.  Pure numpy: no process-group code here.

The QP instances are independent, so the batch is cut into contiguous shards, one process + one
handle + one stream per device, and nothing is exchanged on the data path.  The collectives a
multi-rank CALLER may want around it (barrier, MAX of the elapsed time, gather of the 16-byte
controls) live with that caller: bench_dist.py at the repository root for bench.py and the tests.
"""
from __future__ import annotations


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
