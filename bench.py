#!/usr/bin/env python3
"""bench.py - MPC QP solves/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one pass of the hot path (assembly + ADMM / certified polish, one launch of the solve kernel, which
builds its QP in registers) over one batch of B synthetic controller instances whose inputs are already
resident in HBM.  N=1 workload: config 2 of
BASELINE.json (B=1024 independent initial poses, reference tracking, horizon 30).  With --gpus N
every rank solves its own batch of the same size (weak scaling, no data-path collective; the
instances are independent) and `value` is the whole-job rate.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher environment (RANK / WORLD_SIZE unset) starts the N ranks itself: this
process, which has not touched a GPU yet, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a CHILD and exits with its code.  Under a launcher, WORLD_SIZE must equal --gpus or the run stops.
`--config 5` is BASELINE config 5: 8 192 instances per rank (65 536 over 8 GPUs), config-4 distribution.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "multi-purpose-mpc_amd"), ROOT]

import mpmpc  # noqa: E402
import scenarios  # noqa: E402
import sharding  # noqa: E402
import bench_dist  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md
FP64_VALU_PEAK = 78.6e12   # FLOP/s vector FP64 (spec)


def algorithmic_bytes_per_solve(N, materialised=True):
    """compulsory inputs + outputs per solve, plus the stage-blocked QP written by
    K1 and read by K2 when the two-kernel split materialises it."""
    inp = 8 * (7 * N + 3)
    out = 8 * (5 * N + 3 + 2) + 8
    qp = 2 * 8 * mpmpc.NUM_FIELDS * (N + 1) if materialised else 0
    return inp + out + qp


def k1_bytes_per_solve(N):
    return 8 * (7 * N + 3) + 8 * mpmpc.NUM_FIELDS * (N + 1)


def k2_bytes_per_solve(N):
    # the solve launch assembles its own QP in registers: read wp_id, x0 (3), cc_prev (2N), lb, ub (N each); write
    # z (5N+3), multipliers y (8N+6), u0, residuals, status, iterations.  (The stage-blocked QP is not materialised.)
    return 4 + 8 * (3 + 4 * N) + 8 * (5 * N + 3 + 8 * N + 6 + 2 + 2) + 12


def k2_flops_per_solve(N, admm_iters, ipm_iters):
    """Useful FP64 flops of one solve: per-lane instruction census of the lock-step emulation
    (profiles/census.py: tests/emul with -DMPMPC_COUNT_OPS; FMA = 2, add/mul/div/sqrt = 1) times the
    N+1 lanes that hold a stage.  Fitted per lane as setup + a * ADMM iterations + b * interior-point
    iterations (active-set rounds and certificate amortised into b) at N = 30 and N = 50, linear in N
    between: 2816 + 844 a + 3985 b and 0 + 1434 a + 7097 b (rms error 5 %).  DESIGN.md section 5."""
    t = (N - 30) / 20.0
    c0, c1, c2 = 2816 + t * (0 - 2816), 843.6 + t * (1434 - 843.6), 3985 + t * (7097 - 3985)
    return (N + 1) * (c0 + c1 * admm_iters + c2 * ipm_iters)


def cpu_baseline(tr, sc, seconds=10.0):
    """The oracle's C port (oracle/osqp_port.c: own restatement of the reference's assembly + OSQP +
    certified polish) timed on the host cores of this box, on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    limits = dict(umin=scenarios.UMIN, umax=scenarios.UMAX, xmin=scenarios.XMIN, xmax=scenarios.XMAX,
                  ay_max=scenarios.AY_MAX, wheelbase=scenarios.CAR_LENGTH)
    return oracle_c.timed_baseline(tr, sc, scenarios.WEIGHTS[sc.weights], limits, seconds)


def stock_osqp_leg(tr, sc, ref, seconds=3.0):
    """if the `osqp` package happens to be importable on this box, time the reference-equivalent path
    with it - numpy assembly (oracle/mpc_np.py) + osqp.OSQP().setup(...).solve() per instance at stock defaults,
    one process - and report how far stock OSQP's controls are from the certified optimum.  It is installed
    neither in this image nor on the GPU boxes of this pool, so normally this returns a one-line note."""
    try:
        import osqp  # noqa: F401
    except Exception:
        return {"available": False, "note": "python package `osqp` is not installed on this box"}
    try:
        from scipy import sparse
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mpc_np as M
        otrack = M.Track.sim_track()
        w = M.Weights.time_optimal() if sc.weights == "time_optimal" else M.Weights.stock()
        lim = M.Limits.stock()
        t0, done, worst = time.perf_counter(), 0, 0.0
        for i in range(sc.wp_id.size):
            P, q, A, l, u = M.assemble(otrack, int(sc.wp_id[i]), sc.x0[i], sc.cc_prev[i], sc.lb[i], sc.ub[i], sc.N, w, lim)
            prob = osqp.OSQP()
            prob.setup(P=sparse.csc_matrix(P), q=q, A=sparse.csc_matrix(A), l=l, u=u, verbose=False)
            res = prob.solve()
            done += 1
            if i < ref["status"].size and ref["status"][i] == 1 and res.x is not None and res.x[0] is not None:
                u0 = np.array([res.x[-2 * sc.N], np.arctan(res.x[-2 * sc.N + 1] * scenarios.CAR_LENGTH)])
                worst = max(worst, float(np.max(np.abs(u0 - ref["u0"][i]))))
            if time.perf_counter() - t0 > seconds:
                break
        return {"available": True, "value": done / (time.perf_counter() - t0), "unit": "solves/s", "cores": 1,
                "sample": "%d instances, numpy assembly + stock osqp (defaults) per instance, one process" % done,
                "max_abs_u_minus_uref": worst}
    except Exception as e:          # never let an optional leg break the bench line
        return {"available": True, "note": "stock-osqp leg failed: %r" % (e,)}


def pmc_traffic_bytes(kernel_prefix, B):
    """HBM bytes per launch of one kernel from the committed rocprofv3 PMC summary of THIS command
    (profiles/r1/pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when no summary for this batch size exists."""
    path = os.path.join(ROOT, "profiles", "r1", "pmc_summary.json")
    key_f, key_w = ("pmc_fetch", "pmc_write") if B == 1024 else ("pmc_fetch_b%d" % B, "pmc_write_b%d" % B)
    try:
        d = json.load(open(path))
        f = next(v for k, v in d[key_f].items() if k.startswith(kernel_prefix))["FETCH_SIZE"]["mean"]
        w = next(v for k, v in d[key_w].items() if k.startswith(kernel_prefix))["WRITE_SIZE"]["mean"]
        return (2.0 * f + w) * 1024.0
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="force 64 / 32 / 16 lanes per instance (tuning; 0 = automatic)")
    ap.add_argument("--early-polish", type=int, default=None, help="override the early_polish solver setting")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override any solver setting of the device path (exploration; the oracle keeps its defaults)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        # Start the ranks as children of this process, which has made no GPU call (never re-exec a process that has).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.run(cmd).returncode
        if rc != 0:
            sys.stderr.write("bench.py: the %d-rank run failed (exit code %d)\n" % (args.gpus, rc))
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    dist = None
    if launched:   # launched by torch.distributed.run (also with one rank)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            sys.exit("bench.py: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    tr = scenarios.sim_track()
    spec = scenarios.CONFIGS[args.config]
    B = args.batch or spec.get("B_per_gpu", spec["B"])     # per rank (weak scaling)
    # every rank gets its own slice of a world*B batch drawn from the config's seed
    sc_all = scenarios.make(args.config, tr, B=B * world)
    wp, x0, cc, lb, ub = sharding.shard([sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub], world, rank)
    N = sc_all.N
    Q, R, QN = scenarios.WEIGHTS[sc_all.weights]
    cfg = mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                            scenarios.AY_MAX, scenarios.CAR_LENGTH, circular=True, max_batch=B, device=local_rank)
    overrides = {} if args.early_polish is None else {"early_polish": args.early_polish}
    for kv in args.set:
        k, v = kv.split("=", 1)
        overrides[k] = float(v) if any(c in v for c in ".eE") else int(v)
    settings = mpmpc.default_settings(**overrides)
    h = mpmpc.Handle(cfg, settings)
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_packing(args.lanes)
    h.upload(wp, x0, cc, lb, ub)          # inputs resident in HBM before the timed region

    def barrier():
        h.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        h.solve_resident(B)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.solve_resident(B)
    barrier()
    dt_mine = time.perf_counter() - t0
    dev = "cuda" if dist is not None else None
    dt = bench_dist.max_over_ranks(dist, dt_mine, device=dev)
    dt_ranks = bench_dist.all_ranks(dist, dt_mine, device=dev)

    # per-kernel durations, HIP events on the library's own stream
    reps = max(5, min(args.steps, 20))
    ka, ks = [], []
    for _ in range(reps):
        a, s = h.solve_resident_timed(B)
        ka.append(a)
        ks.append(s)
    ms_k1, ms_k2 = float(np.mean(ka)), float(np.mean(ks))
    sol = h.download(B, want_y=False)

    # one result buffer, outside the timed region: the shards' controls gathered on every rank (the only data
    # collective a caller of the sharded path may want), checked on rank 0 against ONE process solving the whole
    # world*B batch of the same seed on its own GPU
    gather_check = None
    if dist is not None and world > 1:
        u_all, s_all = bench_dist.gather_controls(dist, sol.u0, sol.status, B * world, device=dev)
        if rank == 0:
            cfg1 = mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                                     scenarios.AY_MAX, scenarios.CAR_LENGTH, circular=True, max_batch=B * world,
                                     device=local_rank)
            h1 = mpmpc.Handle(cfg1, settings)
            h1.set_path(tr.kappa, tr.v_ref, tr.ds_next)
            one = h1.solve(sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub)
            h1.close()
            okm = (one.status == 1) & (s_all == 1)
            gather_check = {"instances": int(B * world), "status_equal": bool(np.array_equal(one.status, s_all)),
                            "max_abs_u_diff": float(np.max(np.abs(one.u0[okm] - u_all[okm]))) if okm.any() else None,
                            "note": "all-gathered (u0, status) of the %d shards vs one process solving the whole batch" % world}

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "MPC QP solves/sec (batch, horizon N=%d)" % N,
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "world_size": int(dist.get_world_size()) if dist is not None else 1,
            "ms_per_step_by_rank": [1e3 * t / args.steps for t in dt_ranks],
            "config": {"workload": "config%d: batch=%d independent poses per GPU, %s weights, N=%d, %s corridor, "
                                   "OSQP-default ADMM, certified polish tried after %d iterations" %
                                   (args.config, B, sc_all.weights, N, "obstacle" if sc_all.obstacles else "free",
                                    settings.early_polish),
                       "batch_per_gpu": B, "horizon": N, "parallelism": "batch-shard x%d" % world},
        }
        if gather_check is not None:
            out["gather_check"] = gather_check
        bytes_k2 = k2_bytes_per_solve(N) * B
        out["roofline"] = {"bound": "hbm", "kernel": "mpmpc_solve_kernel", "achieved": bytes_k2 / (ms_k2 * 1e-3) / 1e9,
                           "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_k2 / (ms_k2 * 1e-3) / HBM_PEAK,
                           "traffic": pmc_traffic_bytes("mpmpc_solve_kernel", B) if args.config == 2 else None,
                           "algorithmic_bytes": bytes_k2, "avg_ms": ms_k2,
                           "note": "K2 is FP64-VALU / dependency-chain bound, not HBM bound (DESIGN.md section 5)"}
        flops_k2 = float(np.sum(k2_flops_per_solve(N, sol.iters[:, 0].astype(float), sol.iters[:, 1].astype(float))))
        out["roofline_fp64"] = {"bound": "fp64-valu", "kernel": "mpmpc_solve_kernel",
                                "achieved": flops_k2 / (ms_k2 * 1e-3) / 1e12, "peak": FP64_VALU_PEAK / 1e12,
                                "unit": "TFLOP/s", "frac": flops_k2 / (ms_k2 * 1e-3) / FP64_VALU_PEAK,
                                "flops_per_solve_mean": flops_k2 / B,
                                "note": "useful flops (N+1 of an instance's lanes hold a stage); the slowest wave sets the time"}
        bytes_k1 = k1_bytes_per_solve(N) * B
        out["roofline_assembly"] = {"bound": "hbm", "kernel": "mpmpc_assemble_kernel",
                                    "achieved": bytes_k1 / (ms_k1 * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                                    "unit": "GB/s", "frac": bytes_k1 / (ms_k1 * 1e-3) / HBM_PEAK, "avg_ms": ms_k1,
                                    "traffic": pmc_traffic_bytes("mpmpc_assemble_kernel", B) if args.config == 2 else None,
                                    "algorithmic_bytes": bytes_k1}
        st, cnt = np.unique(sol.status, return_counts=True)
        out["status_counts"] = {int(s): int(c) for s, c in zip(st, cnt)}
        out["iters"] = {"admm_mean": float(sol.iters[:, 0].mean()), "admm_max": int(sol.iters[:, 0].max()),
                        "ipm_mean": float(sol.iters[:, 1].mean()), "ipm_max": int(sol.iters[:, 1].max()),
                        "ipm_histogram": {int(k): int(v) for k, v in zip(*np.unique(sol.iters[:, 1], return_counts=True))}}
        # the same step from HOST buffers (mpmpc_solve: upload + assembly and solve in one launch + download): reported, never `value`
        t1 = time.perf_counter()
        for _ in range(5):
            h.solve(wp, x0, cc, lb, ub)
        out["host_buffers"] = {"value": 5 * B / (time.perf_counter() - t1), "unit": "solves/s",
                               "note": "PCIe-inclusive rate of mpmpc_solve on one GPU (pageable numpy buffers)"}
        if not args.no_cpu:
            sc_rank = scenarios.Scenario(sc_all.name, N, sc_all.weights, sc_all.obstacles, wp, x0, cc, lb, ub)
            base, ref = cpu_baseline(tr, sc_rank)
            out["cpu_baseline"] = base
            ns = ref["status"].size
            both = (ref["status"] == 1) & (sol.status[:ns] == 1)
            out["max_abs_u_minus_uref"] = float(np.max(np.abs(sol.u0[:ns][both] - ref["u0"][both]))) if both.any() else None
            out["status_agreement"] = float(np.mean(ref["status"] == sol.status[:ns]))
            # whole plan (z without the cost-free kappa_{N-1} and e_psi_N) against the certified optimum
            keep = np.ones(5 * N + 3, bool)
            keep[[3 * N + 1, 5 * N + 2]] = False
            out["max_abs_plan_minus_ref"] = (float(np.max(np.abs(sol.z[:ns][both][:, keep] - ref["z"][both][:, keep])))
                                            if both.any() else None)
            # what stock settings deliver: OSQP defaults (eps 1e-3, no polish) on the device against the same optimum
            h.set_settings(mpmpc.default_settings(polish=0, early_polish=0))
            stock = h.solve(wp[:ns], x0[:ns], cc[:ns], lb[:ns], ub[:ns])
            ok = both & ((stock.status == 1) | (stock.status == 2))
            out["stock_osqp_settings"] = {
                "max_abs_u_minus_uref": float(np.max(np.abs(stock.u0[ok] - ref["u0"][ok]))) if ok.any() else None,
                "admm_iters_mean": float(stock.iters[:, 0].mean()), "admm_iters_max": int(stock.iters[:, 0].max()),
                "note": "device run at OSQP's defaults (eps_abs = eps_rel = 1e-3, no polish) vs the certified optimum"}
            out["stock_osqp_package"] = stock_osqp_leg(tr, sc_rank, ref)
            out["parity_sample"] = int(ns)
            out["host_cores"] = os.cpu_count()
            out["host_cores_usable"] = base.get("usable_cpus")
        print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
