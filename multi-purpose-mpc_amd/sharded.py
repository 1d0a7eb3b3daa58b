"""Product-level multi-device entry point: ONE process drives every GPU of the node (contiguous shards, one
handle + its streams per device, no collective on the data path - the instances are independent).

    smpc = ShardedBatchMPC(car, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, max_batch=65536)   # all visible GPUs
    u, plan, status = smpc.get_control_batch(wp_id, x0, cc_prev, lb, ub)

serves the loop of the reference's driver (src/simulation.py:134-140: u = mpc.get_control(); car.drive(u)) for a fleet spread
over the node without torch, torchrun or a process group: every call cuts the batch with sharding.shard_bounds, STARTS the
shard of every device (upload + launch are asynchronous on the handle's stream), and only then collects the results in shard
order - the devices work side by side, the host thread never waits for one before it has fed the next.  bench.py --gpus N
--single-process times exactly this class; the one-process-per-GPU path (torch.distributed / RCCL barrier) stays what the
driver's contract asks for.
"""
from __future__ import annotations

import numpy as np

import mpmpc
import sharding


class ShardedHandles:
    """The mechanism, free of the MPC classes: `handles` are mpmpc.Handle objects (one per device, or any object with the same
    upload / solve_resident / download - or just solve - surface: the CPU tests pass emulation backends)."""

    def __init__(self, handles):
        if not handles:
            raise ValueError("need at least one handle")
        self.handles = list(handles)

    @property
    def world(self):
        return len(self.handles)

    def bounds(self, total):
        return [sharding.shard_bounds(total, self.world, r) for r in range(self.world)]

    def set_path(self, kappa, v_ref, ds_next):
        for h in self.handles:
            h.set_path(kappa, v_ref, ds_next)

    def set_corridor(self, ub, lb):
        for h in self.handles:
            h.set_corridor(ub, lb)

    def solve(self, wp_id, x0, cc_prev, lb=None, ub=None, want_y=False) -> mpmpc.Solution:
        """All shards started, then all collected; -> one Solution in the caller's instance order."""
        wp_id = np.ascontiguousarray(wp_id, np.int32)
        arrays = [wp_id, np.ascontiguousarray(x0, float), np.ascontiguousarray(cc_prev, float)]
        rows = lb is not None
        if rows:
            arrays += [np.ascontiguousarray(lb, float), np.ascontiguousarray(ub, float)]
        B = wp_id.size
        started = []
        for r, (h, (lo, hi)) in enumerate(zip(self.handles, self.bounds(B))):
            if hi == lo:
                started.append(None)
                continue
            part = [a[lo:hi] for a in arrays] + ([] if rows else [None, None])
            if hasattr(h, "upload") and hasattr(h, "solve_resident"):
                if hasattr(h, "set_outputs"):
                    h.set_outputs(want_y)
                h.upload(*part)
                h.solve_resident(hi - lo)          # asynchronous: the next device is fed while this one works
                started.append(("device", hi - lo))
            else:
                started.append(("call", part))
        parts = []
        for h, st in zip(self.handles, started):
            if st is None:
                continue
            parts.append(h.download(st[1], want_y=want_y) if st[0] == "device" else h.solve(*st[1], want_y=want_y))
        cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
        return mpmpc.Solution(cat("z"), cat("u0"), cat("status"), cat("iters"), cat("resid"), cat("y") if want_y else None)

    def close(self):
        for h in self.handles:
            if hasattr(h, "close"):
                h.close()


class ShardedBatchMPC:
    """BatchMPC over several devices: same constructor arguments plus `devices` (HIP ordinals; default: every visible one),
    same get_control_batch.  max_batch is the WHOLE fleet; each device's handle is sized for its shard."""

    def __init__(self, model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, max_batch, settings=None, devices=None,
                 corridor=None):
        from MPC import _make_config
        self.N, self.model = N, model
        self.settings = settings or mpmpc.default_settings()
        if devices is None:
            devices = list(range(mpmpc.device_count()))
        if not devices:
            raise mpmpc.MpmpcError("no HIP device visible (this library has no CPU fallback)")
        world = len(devices)
        per = -(-int(max_batch) // world)
        hs = []
        for d in devices:
            cfg = _make_config(model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, per, int(d))
            hs.append(mpmpc.Handle(cfg, self.settings))
        self.shards = ShardedHandles(hs)
        kappa, v_ref, ds = model.reference_path.tables()
        if np.any(np.isnan(v_ref)):
            raise RuntimeError("reference path has no speed profile (call compute_speed_profile first)")
        self.shards.set_path(kappa, v_ref, ds)
        if corridor is not None:
            self.shards.set_corridor(*corridor)

    def get_control_batch(self, wp_id, x0, cc_prev, lb=None, ub=None):
        """-> (u [B,2] = (v, delta), plan [B,2N] with delta entries, status [B], Solution)."""
        sol = self.shards.solve(wp_id, x0, cc_prev, lb, ub)
        plan = sol.z[:, -2 * self.N:].copy()
        plan[:, 1::2] = np.arctan(plan[:, 1::2] * self.model.length)
        return sol.u0, plan, sol.status, sol

    def close(self):
        self.shards.close()
