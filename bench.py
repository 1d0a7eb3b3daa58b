#!/usr/bin/env python3
"""bench.py - MPC QP solves/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one pass of the hot path over one batch of B synthetic controller instances whose inputs are already resident
in HBM: the solve launch assembles its QP in registers (the reference's MPC._init_problem) and solves it.  With the
reference's own weights that is the reduced-native kernel (closed-form speed, Ruiz pass, interior point from x = 0,
active-set round, KKT certificate on the (e_y, e_psi, kappa) problem) plus a tail launch of the general
kernel for what it cannot certify (phase 1 / Farkas ray; the OSQP ADMM iteration only as the last fallback).  N=1 workload: config 2 of
BASELINE.json (B=1024 independent initial poses, reference tracking, horizon 30).  With --gpus N
every rank solves its own batch of the same size (weak scaling, no data-path collective; the
instances are independent) and `value` is the whole-job rate.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N --single-process        # ONE process drives the N devices (no torch, no process group): same line schema
  python bench.py --gpus N [--single-process] --dry-run   # rehearsal of either multi-GPU path on a box without GPUs (value: null)
  ... --no-extra-legs                               # under a profiler: the timed loop's kernels only

Beside `value` (median of >= 25 regions of exactly K steps, four resident launches in flight) the line carries `value_200_step_regions`
/ `ramp_drain_share` (the same loop in long regions), `value_one_launch_in_flight` (a caller whose step k + 1 needs step k),
`admm_stock_mode` (the restated OSQP alone), `roofline` (HBM; + `roofline_fp64`, `roofline_assembly`), `cpu_baseline` (+ `_stock`).

`--gpus N` with N > 1 and no launcher environment (RANK / WORLD_SIZE unset) starts the N ranks itself: this
process, which has not touched a GPU yet, runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a CHILD and exits with its code.  Under a launcher, WORLD_SIZE must equal --gpus or the run stops.
`--config 5` is BASELINE config 5: 8 192 instances per rank (65 536 over 8 GPUs), config-4 distribution.
"""
from __future__ import annotations

import os as _os
# Four resident launches in flight (--pipeline 4) need four hardware queues of their own: the HIP runtime's default of 4
# (one of them taken) makes two of the handle's streams share a queue, which serialises them (profiles/r4/depth_sweep.txt).
# Read by the runtime at its first call in this process; an operator's own setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
HBM_ACHIEVABLE = 6.29e12   # B/s, the guide's measured-achievable rate (79 % of the peak): a streaming kernel cannot beat it from HBM
PROFILE_ROUNDS = ("r6", "r5", "r4")      # committed rocprofv3 summaries are looked up newest round first (each names its library sources)
FP64_VALU_PEAK = 78.6e12   # FLOP/s vector FP64 (spec)


def algorithmic_bytes_per_solve(N):
    """SURVEY.md 8(d), compulsory traffic of one solve, independent of the kernel design:
    inputs  x0 (3) + per stage (kappa, v_ref, ds, lb, ub, cc_prev[2]) = 8 (7N + 3) B,
    outputs z (5N + 3) + u0 (2) doubles + status / iterations 8 B    = 8 (5N + 5) + 8 B:   2 952 B at N = 30."""
    return 8 * (7 * N + 3) + 8 * (5 * N + 3 + 2) + 8


def k1_bytes_per_solve(N):
    """the stand-alone assembly kernel (mpmpc_assemble, off the solve path): inputs + the stage-blocked QP it writes"""
    return 8 * (7 * N + 3) + 8 * mpmpc.NUM_FIELDS * (N + 1)


# FP64 flops of one solve at the default settings, fitted on the instruction census of the lock-step emulation of the
# SAME lane code (profiles/census.py; FMA = 2, other arithmetic = 1, compares / selects / lane moves = 0) as
#     flops = c0 + c1 * interior-point iterations      (rms error below 2 %: one active-set round almost always)
# "algorithmic": the structure-exploiting count of the implemented recurrence - lane-parallel instructions
#   count for the lanes that hold a stage, every serial sweep of the twisted factorisation / substitutions counts ONE
#   step per stage (per stage: factor 180 flops, KKT solve 110).
# "executed": every wave instruction times the N + 1 stage-holding lanes, serial steps as often as they are executed
#   (round 1's figure).
# key: (N + 1 <= 32, i.e. one lane per stage with the split interior-point layout;  reduced polish (mpmpc_settings::reduce
#       applies: t carries neither cost nor bound);  certified optimal / Farkas-certified)
_FLOPS = {
    (True, True, 1): dict(algorithmic=(40708.0, 18134.0), executed=(98084.0, 40982.0), N=30),
    (True, True, -3): dict(algorithmic=(48823.0, 22075.0), executed=(80271.0, 49912.0), N=30),
    (True, False, 1): dict(algorithmic=(60242.0, 34985.0), executed=(207452.0, 97650.0), N=30),
    (True, False, -3): dict(algorithmic=(59657.0, 41835.0), executed=(141472.0, 116563.0), N=30),
    (False, False, 1): dict(algorithmic=(105506.0, 46289.0), executed=(544228.0, 246133.0), N=50),
}


# the reduced-native kernels (mpmpc_settings::native; profiles/census.py through the launcher's sequence of kernels)
# (cyclic-reduction factorisation of the 16-lane chains: every stage is eliminated at one of the four levels, so a level's
#  work counts once per stage in "algorithmic" - like one step per stage of a serial sweep - and four times in "executed")
# the terminal-time kernels (config 3, N = 50; profiles/census.py 3): the active-set rounds vary too much for an intercept, the
# fit is through the origin (round 5: mean 0.451 / 0.944 MFLOP per solve at 9.7 iterations)
_FLOPS_NATIVE_TT = {1: dict(algorithmic=(0.0, 46500.0), executed=(0.0, 97300.0), N=50)}
_FLOPS_NATIVE = {
    # (refitted on the final round-5 code, profiles/census.py 2 256 / 4 512: the zero half of the last cyclic-reduction level
    #  is no longer formed - and no longer counted; infeasible instances through the reduced-native tail solver, two per
    #  wave, whose phase 1 now runs to its converged optimum inside the band: more iterations, hence the intercepts)
    1: dict(algorithmic=(30484.0, 18553.0), executed=(53147.0, 27954.0), N=30),
    -3: dict(algorithmic=(-1566.0, 24631.0), executed=(-34128.0, 42580.0), N=30),
}


def qp_objective(qp, N, z):
    """1/2 z'Pz + q'z per instance from the stage-blocked fields of mpmpc_assemble (diagonal weights: fields 22-26 the
    cost diagonal, 17-21 the cost vector; include/mpmpc.h)"""
    B = qp.shape[1]
    f = qp[:, :, :N + 1]
    ne = 3 * (N + 1)
    x, u = z[:, :ne].reshape(B, N + 1, 3), z[:, ne:].reshape(B, N, 2)
    Px, qx = np.moveaxis(f[22:25], 0, -1), np.moveaxis(f[17:20], 0, -1)
    Pu, qu = np.moveaxis(f[25:27, :, :N], 0, -1), np.moveaxis(f[20:22, :, :N], 0, -1)
    return (0.5 * Px * x * x + qx * x).sum(axis=(1, 2)) + (0.5 * Pu * u * u + qu * u).sum(axis=(1, 2))


def algorithm_text(cfg, settings):
    """what the solve launches of this configuration execute, in words (for config.workload)"""
    if not settings.polish:
        return "restated OSQP ADMM at the settings given (no polish): the reference's own solver call"
    if native_tt_path(cfg, settings):
        return ("terminal-time reduced-native kernel, one instance per wave: t eliminated (t_N is a linear functional of e_y and v), one Ruiz "
                "pass, interior point from x = 0 on the (e_y, e_psi, kappa, v) QP with its rank-one time term (2x2-block cyclic-reduction "
                "Cholesky + Sherman-Morrison), active-set round(s), KKT certificate, roll-forward of t; uncertified instances go to a tail "
                "launch of the general kernel (phase 1 / Farkas ray, then the OSQP ADMM iteration)")
    if native_path(cfg, settings):
        return ("reduced-native kernel per instance: speed in closed form, one Ruiz pass, interior point from x = 0 (no OSQP iterate "
                "is computed: iters[:, 0] = 1 marks the attempt), active-set round(s), KKT certificate on the (e_y, e_psi, kappa) QP, roll-forward of t; "
                "uncertified instances go to a tail launch of the reduced-native tail kernel (two instances per wave for horizons up to 31, "
                "two waves per SIMD like the first kernel): phase 1 (Farkas ray / least-violation point) and one more attempt; only what that cannot decide goes "
                "on to the general kernel and the OSQP ADMM iteration (each tail launch is enqueued with the step while the launches "
                "seen so far leave something for it, otherwise only when a launch turns out to need it - checked at every sync, inside "
                "the timed region)")
    red = "reduced 2x2-block" if reduced_polish(cfg, settings) else "full 3x3-block"
    return ("general kernel per instance: %d Ruiz pass(es), %s, %s interior point + active-set round(s) + KKT "
            "certificate; phase 1 (Farkas ray) and then the full OSQP ADMM run for what that cannot certify" %
            (settings.early_scaling, ("%d OSQP start step(s)" % settings.early_polish) if settings.early_start else
             "start from x = 0 (no OSQP iterate is computed: iters[:, 0] = 1 marks the attempt)", red))


def native_tt_path(cfg, settings):
    """mirror of mpmpc::reduced_native_tt (csrc/mpmpc_reduced_t.hpp): terminal cost on the time state and nothing else on it"""
    return bool(settings.native and settings.reduce and settings.polish and cfg.Q[2] == 0.0 and cfg.QN[2] > 0.0 and not any(cfg.QN_offdiag)
                and cfg.R[0] > 0.0 and cfg.xmin[2] <= -1e30 and cfg.xmax[2] >= 1e30 and cfg.xmin[1] <= -1e30 and cfg.xmax[1] >= 1e30
                and settings.early_polish == 1 and settings.max_iter > 1 and settings.ipm_start_mu > 0.0 and settings.scaling > 0)


def native_path(cfg, settings):
    """mirror of mpmpc::reduced_native (csrc/mpmpc_reduced.hpp): do the batch launches run the reduced-native kernels?"""
    return bool(settings.native and reduced_polish(cfg, settings) and settings.early_polish == 1 and settings.max_iter > 1
                and settings.ipm_start_mu > 0.0 and settings.scaling > 0)


def rocprof_kernel_average(config, B, lib_version, steps=None):
    """Average duration [ms] of a step's solve launches in the committed `rocprofv3 --kernel-trace` of this command, over the
    TIMED launches only (profiles/r4/kernel_timed.json, written by profiles/summarize.py from the trace rows: the prewarm /
    warm-up launches and the profile launches after the timed loop are dropped) - reported when, and only when, that trace
    was taken on a library built from the same sources as the one running now.  -> (ms or None, source text)"""
    name = {(2, 1024): "bench_cfg2", (2, 65536): "bench_cfg2_b65536", (3, 4096): "bench_cfg3", (4, 8192): "bench_cfg4", (5, 8192): "bench_cfg5"}.get((config, B))
    d, rnd = None, None
    for rnd in PROFILE_ROUNDS:
        try:
            allk = json.load(open(os.path.join(ROOT, "profiles", rnd, "kernel_timed.json")))
            d = allk[name]
            # (config 2 is traced twice: with the defaults and with the driver's own command - the one with this run's K is the "same command")
            alt = allk.get(name + "_driver_command")
            if alt and steps is not None and int(alt.get("steps", -1)) == int(steps) != int(d.get("steps", -1)):
                d = alt
            break
        except Exception:
            d = None
    if d is None:
        return None, "no kernel trace committed for this workload (profiles/%s/kernel_timed.json)" % PROFILE_ROUNDS[0]
    src = (d.get("library") or "").split("src ")[-1].rstrip(")")
    if not src or src not in lib_version:
        return None, "profiles/%s/kernel_timed.json was measured on library sources %s, this run is %s" % (rnd, src or "?", lib_version)
    ms = sum(k["avg_us"] * k["launches"] for k in d["kernels"].values()) / d["launches"] * 1e-3
    return ms, ("profiles/%s/kernel_timed.json (rocprofv3 --kernel-trace of this command, same library sources %s): the %d timed launches, "
                "kernels of a step summed; %.2f solve kernels on the chip on average over the window" % (rnd, src, d["launches"], d["solve_kernels_in_flight_avg"]))


def reduced_polish(cfg, settings):
    """mirror of mpmpc::reducible (csrc/mpmpc_core.hpp): may the polish solve the (e_y, e_psi, kappa) problem?"""
    return bool(settings.reduce and settings.polish and cfg.Q[2] == 0.0 and cfg.QN[2] == 0.0 and not any(cfg.QN_offdiag)
                and cfg.R[0] > 0.0 and cfg.xmin[2] <= -1e30 and cfg.xmax[2] >= 1e30 and cfg.xmin[1] <= -1e30 and cfg.xmax[1] >= 1e30)


def k2_flops(N, status, ipm_iters, kind, reduced, native=False, native_tt=False):
    """flops of a batch by the fit above, scaled linearly in the number of stages away from the fitted horizon"""
    total = 0.0
    split = N + 1 <= 32
    for stt in (1, -3):
        m = (status == stt) if stt == -3 else (status != -3)
        if not m.any():
            continue
        f = _FLOPS.get((split, reduced, stt)) or _FLOPS.get((split, False, stt)) or _FLOPS[(split, False, 1)]
        if native:
            f = _FLOPS_NATIVE[stt]
        if native_tt and stt == 1:
            f = _FLOPS_NATIVE_TT[stt]
        c0, c1 = f[kind]
        total += float(np.sum(c0 + c1 * ipm_iters[m].astype(float))) * (N + 1) / (f["N"] + 1)
    return total


def cpu_baseline(tr, sc, seconds=10.0, stock=False):
    """The oracle's C port (oracle/osqp_port.c: own restatement of the reference's assembly + OSQP +
    certified polish) timed on the host cores of this box, on a bounded sample of the workload.
    stock: the port at polish = 0, early_polish = 0, phase1 = 0 - restated OSQP at its defaults and nothing else, the
    arithmetic of the reference's own solver call (src/MPC.py:159,183)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    limits = dict(umin=scenarios.UMIN, umax=scenarios.UMAX, xmin=scenarios.XMIN, xmax=scenarios.XMAX,
                  ay_max=scenarios.AY_MAX, wheelbase=scenarios.CAR_LENGTH)
    return oracle_c.timed_baseline(tr, sc, scenarios.WEIGHTS[sc.weights], limits, seconds, stock=stock)


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


def pmc_traffic_bytes(kernel_prefix, B, lib_version, config=2):
    """HBM bytes per launch of one kernel from the committed rocprofv3 PMC summary of THIS command
    (profiles/r4/pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  The summary records the source hash of the library it was measured
    on: the figure is only reported when the library running now was built from the same sources, otherwise None
    (with the reason) - a counter reading of another kernel is not this run's traffic."""
    key_f, key_w = ("pmc_fetch", "pmc_write") if B == 1024 else ("pmc_fetch_b%d" % B, "pmc_write_b%d" % B)
    if config != 2:
        key_f, key_w = "pmc_fetch_cfg%d" % abs(config), "pmc_write_cfg%d" % abs(config)
    d, why = None, "no PMC summary committed (profiles/%s/pmc_summary.json)" % PROFILE_ROUNDS[0]
    for rnd in PROFILE_ROUNDS:
        try:
            cand = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")))
        except Exception:
            continue
        src = cand.get("library_source_hash", "")
        if src and src in lib_version:
            d = cand
            break
        why = "profiles/%s/pmc_summary.json was measured on library sources %s, this run is %s" % (rnd, src or "?", lib_version)
    if d is None:
        return None, why
    try:
        f = next(v for k, v in d[key_f].items() if k.startswith(kernel_prefix))["FETCH_SIZE"]["mean"]
        w = next(v for k, v in d[key_w].items() if k.startswith(kernel_prefix))["WRITE_SIZE"]["mean"]
        note = ""
        # a launch that leaves a tail also runs its tail kernel(s): their bytes belong to the launch (one tail launch per launch)
        # (the EXACT instantiations the launcher uses for a tail - not any kernel of the family: a pass that also ran the restated
        #  OSQP at stock settings holds mpmpc_solve_kernel<.., 3> launches of another kind beside the tail's <.., 2>; ADVICE r5.
        #  The profiling scripts now pass --no-extra-legs, so such launches are not in new summaries at all.)
        tails = ("mpmpc_reduced_tail_kernel<", "mpmpc_solve_kernel<64, 16, false, 2>", "mpmpc_solve_kernel<64, 32, false, 2>") if config != 3 \
            else ("mpmpc_solve_kernel<64, 32, false, 3>",)
        for tk in tails:
            if tk.startswith(kernel_prefix) or config == 2 or config < 0:          # (config < 0: one kernel alone, K1)
                continue
            tf = [v["FETCH_SIZE"]["mean"] for k, v in d[key_f].items() if k.startswith(tk)]
            tw = [v["WRITE_SIZE"]["mean"] for k, v in d[key_w].items() if k.startswith(tk)]
            if len(tf) == 1 and len(tw) == 1:
                f, w, note = f + tf[0], w + tw[0], note + " + " + tk.rstrip("<")
        return (2.0 * f + w) * 1024.0, "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of %s%s, same library sources %s)" % (
            rnd, kernel_prefix, note, src)
    except Exception:
        return None, "profiles/%s/pmc_summary.json has no counters for this kernel at B = %d" % (rnd, B)


def check_ranks(args, dist, rank, world, ranks):
    """Every rank holds the same per-rank reports (bench_dist.rank_reports all-gathers them): all of them leave, non-zero and
    before rank 0 prints, unless the process group has --gpus ranks and every rank drives a device of its own."""
    if world <= 1:
        return
    devs = [r_["device"] for r_ in ranks] if ranks else []
    if dist is None or int(dist.get_world_size()) != args.gpus or len(set(devs)) != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d, but the process group has %s ranks on devices %s: no line printed\n"
                             % (args.gpus, dist.get_world_size() if dist is not None else None, devs))
        sys.exit(3)


def single_process_bench(args, real_stdout):
    """--gpus N --single-process: the N shards of a world x B batch on N handles of THIS process (device r for shard r; no
    torch, no process group, no collective) - a second way to run config 5 should the box's torch / RCCL pairing misbehave.
    A step = one resident launch on every handle; the clock stops when every device is idle.  Same schema as the torchrun
    line where it applies: `ranks` (device and status counts per handle), `roofline`, `value_one_launch_in_flight`,
    `gather_check`.  With --dry-run the handles are the CPU emulation's stand-ins (a rehearsal: value null)."""
    import sharded
    tr = scenarios.sim_track()
    spec = scenarios.CONFIGS[args.config]
    B = args.batch or spec.get("B_per_gpu", spec["B"])
    world = args.gpus
    if args.dry_run:
        sys.path.insert(0, os.path.join(ROOT, "bench_support"))
        import emulation
        n_dev = world
    else:
        n_dev = mpmpc.device_count()
        if n_dev < 1:
            sys.exit("bench.py: no HIP device visible")
        if world > n_dev and not os.environ.get("MPMPC_BENCH_SHARE_DEVICE"):
            sys.exit("bench.py: --gpus %d --single-process but %d device(s) visible (MPMPC_BENCH_SHARE_DEVICE=1 puts the handles on "
                     "device 0 for a functional check)" % (world, n_dev))
    sc_all = scenarios.make(args.config, tr, B=B * world)
    N = sc_all.N
    Q, R, QN = scenarios.WEIGHTS[sc_all.weights]
    settings = mpmpc.default_settings()
    hs = []
    for r in range(world):
        cfg = mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX,
                                scenarios.CAR_LENGTH, circular=True, max_batch=B, device=r if world <= n_dev else 0)
        h = emulation.DryHandle(cfg, settings) if args.dry_run else mpmpc.Handle(cfg, settings)
        h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        h.set_outputs(want_y=False)
        h.set_pipeline(args.pipeline)
        h.upload(*sharding.shard([sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub], world, r))
        hs.append(h)

    def step():
        for h in hs:
            h.solve_resident(B)

    def sync():
        for h in hs:
            h.sync()

    def timed(repeats):
        dts = []
        for _ in range(max(1, repeats)):
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            dts.append(time.perf_counter() - t0)
        return dts
    for _ in range((0 if args.dry_run else args.prewarm) + args.warmup):
        step()
    dts = timed(args.repeats)
    dt = float(np.median(dts))
    # per-device status counts of the timed launches' results (the torchrun line's `ranks`)
    ranks = []
    for r, h in enumerate(hs):
        st = h.download(B, want_y=False).status
        ranks.append({"rank": r, "device": int(h.cfg.device), "solved": int(np.sum(st == 1)), "solved_inaccurate": int(np.sum(st == 2)),
                      "infeasible": int(np.sum(st == -3)), "other": int(np.sum((st != 1) & (st != 2) & (st != -3)))})
    # the other operating point: one launch in flight per handle
    for h in hs:
        h.set_pipeline(1)
    for _ in range(max(args.warmup, 5)):
        step()
    dt1 = float(np.median(timed(args.repeats)))
    for h in hs:
        h.set_pipeline(args.pipeline)
    # launch durations on device 0's streams (HIP events around each launch of the pipelined pattern)
    roof = None
    if not args.dry_run:
        n_prof = max(8, min(args.steps, 64))
        each, span = hs[0].solve_resident_profile(B, n_prof)
        bytes_k2 = algorithmic_bytes_per_solve(N) * B
        agg = bytes_k2 * world * args.steps / dt
        roof = {"bound": "hbm", "achieved": agg / 1e9, "peak": world * HBM_PEAK / 1e9, "unit": "GB/s", "frac": agg / (world * HBM_PEAK),
                "frac_per_kernel": bytes_k2 / (float(np.mean(each)) * 1e-3) / HBM_PEAK, "avg_ms": float(np.mean(each)), "span_ms": float(span),
                "launches_measured": int(n_prof), "algorithmic_bytes": bytes_k2, "bytes_per_solve": algorithmic_bytes_per_solve(N), "traffic": None,
                "note": "achieved = algorithmic bytes of all devices' timed launches / the timed region, peak = n_gpus x 8 TB/s; avg_ms / "
                        "frac_per_kernel: HIP events around each launch on device 0's own streams (mpmpc_solve_resident_profile).  The solve "
                        "kernels are FP64-VALU bound, not HBM bound: see the default run's roofline_fp64"}
    # the shards through the product class against one handle solving the whole batch (outside the timed region)
    sh = sharded.ShardedHandles(hs)
    got = sh.solve(sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub)
    cfg1 = mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX,
                             scenarios.CAR_LENGTH, circular=True, max_batch=B * world, device=0)
    h1 = emulation.DryHandle(cfg1, settings) if args.dry_run else mpmpc.Handle(cfg1, settings)
    h1.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    one = h1.solve(sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub)
    h1.close()
    out = {"metric": "MPC QP solves/sec (batch, horizon N=%d)" % N, "value": None if args.dry_run else world * B * args.steps / dt, "unit": "solves/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None if args.dry_run else 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "repeats": len(dts), "ms_per_step_min": None if args.dry_run else 1e3 * min(dts) / args.steps,
           "ms_per_step_max": None if args.dry_run else 1e3 * max(dts) / args.steps,
           "value_one_launch_in_flight": None if args.dry_run else world * B * args.steps / dt1,
           "one_launch_in_flight": {"value": None if args.dry_run else world * B * args.steps / dt1, "unit": "solves/s",
                                    "ms_per_step": None if args.dry_run else 1e3 * dt1 / args.steps,
                                    "note": "mpmpc_set_pipeline(h, 1) on every handle: same K, same repeats as `value`"},
           "config": {"workload": "config%d: batch=%d independent poses per GPU, %s weights, N=%d; ONE process, one handle per device "
                                  "(sharded.ShardedHandles), no process group%s" % (args.config, B, sc_all.weights, N, "; REHEARSAL on the CPU emulation" if args.dry_run else ""),
                      "batch_per_gpu": B, "horizon": N, "parallelism": "single-process batch-shard x%d" % world},
           "devices": [int(h.cfg.device) for h in hs], "ranks": ranks, "rccl_world_size": 0, "launches_in_flight": int(args.pipeline),
           "gather_check": {"same_status": bool(np.array_equal(got.status, one.status)), "same_u0": bool(np.array_equal(got.u0, one.u0)),
                            "instances": int(B * world)},
           "library": hs[0].lib.mpmpc_version().decode()}
    if roof:
        out["roofline"] = roof
    if args.dry_run:
        out["dry_run"] = True
        out["note"] = "handles, shards, status counts and the gather check rehearsed on the kernels' CPU emulation; not a measurement"
    # what --gpus promised: that many handles, a device each (MPMPC_BENCH_SHARE_DEVICE=1: a functional run on one device, said so)
    shared = bool(os.environ.get("MPMPC_BENCH_SHARE_DEVICE")) and not args.dry_run
    if len(set(out["devices"])) != world and not shared:
        sys.stderr.write("bench.py: --gpus %d --single-process, but the handles sit on devices %s: no line printed\n" % (world, out["devices"]))
        sys.exit(3)
    if shared:
        out["devices_shared"] = True
    real_stdout.write(json.dumps(out) + "\n")
    real_stdout.flush()
    sh.close()


def main():
    # Only the JSON line may reach stdout: libraries loaded below print banners there (RCCL its version block at
    # communicator creation).  File descriptor 1 points at stderr for the duration of the run; the line is written to
    # the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    try:
        _main(real_stdout)
    finally:
        real_stdout.flush()


def _main(real_stdout):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--pipelined", action="store_true",
                    help="also measure two batches in flight (two handles fed in turn); off by default so that a profile of the "
                         "default command holds the launches of the timed loop only")
    ap.add_argument("--prewarm", type=int, default=300, help="untimed launches before the W warm-up steps (clock ramp)")
    ap.add_argument("--repeats", type=int, default=25, help="the timed region of K steps is run this many times; value = median repeat")
    ap.add_argument("--pipeline", type=int, default=4, help="resident launches in flight inside the handle (mpmpc_set_pipeline): 1 .. 8")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the legs beside the timed loop that launch OTHER kernels or packings (one launch in flight, the restated "
                         "OSQP at stock settings): what the profiling scripts pass, so that a kernel trace / PMC pass of the command "
                         "holds the kernels of the timed loop only (ADVICE r5)")
    ap.add_argument("--single-process", action="store_true",
                    help="with --gpus N: ONE process drives the N devices (one handle per device, multi-purpose-mpc_amd/sharded.py) "
                         "instead of one rank per GPU under torch.distributed; a second way to run config 5")
    ap.add_argument("--lanes", type=int, default=0, help="force 64 / 32 / 16 lanes per instance (tuning; 0 = automatic)")
    ap.add_argument("--early-polish", type=int, default=None, help="override the early_polish solver setting")
    ap.add_argument("--dry-run", action="store_true",
                    help="REHEARSAL of the multi-rank plumbing on a box without GPUs (tests/test_distributed_cpu.py): the launcher, the "
                         "rank -> device mapping, the sharding, the barriers / MAX reductions and the gather check run as they will on "
                         "8 GPUs, over gloo instead of RCCL and with the kernels' CPU emulation (tests/) in place of libmpmpc.so.  The "
                         "line it prints carries dry_run: true and value: null - nothing in it is a measurement")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override any solver setting of the device path (exploration; the oracle keeps its defaults)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.single_process:
        return single_process_bench(args, real_stdout)
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
        # (the ranks inherit this process's environment: GPU_MAX_HW_QUEUES=8 set above reaches every one of them)
        rc = subprocess.run(cmd, stdout=real_stdout).returncode      # rank 0's JSON line goes to OUR stdout
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
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
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
    if args.dry_run:
        sys.path.insert(0, os.path.join(ROOT, "bench_support"))
        import emulation
        h = emulation.DryHandle(cfg, settings)          # the handle's resident surface on the CPU emulation (rehearsal infrastructure)
    else:
        h = mpmpc.Handle(cfg, settings)
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_packing(args.lanes)
    h.set_outputs(want_y=False)          # the step needs (u0, z, status): no multipliers are stored
    h.set_pipeline(args.pipeline)
    h.upload(wp, x0, cc, lb, ub)          # inputs resident in HBM before the timed region

    def barrier():
        h.sync()
        if dist is not None:
            if not args.dry_run:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    # Clock ramp, outside everything: the chip needs a few milliseconds of load to reach its sustained clocks, and the
    # W warm-up steps of a short run (W = 2 steps of 0.08 ms) are over before it has.  Untimed, like the upload.
    n_prewarm = 0
    for _ in range(args.prewarm):
        h.solve_resident(B)
        n_prewarm += 1
    barrier()
    # ... and, after a long idle stretch (the host-side set-up of config 5 takes seconds), a fixed number of launches is
    # over before the clocks are back: measured once as 524 -> 297 us per launch over the first 400 TIMED launches of a
    # config-5 run (profiles/README.md).  So the ramp goes on in groups of >= 10 ms until a group is no faster than the best
    # before it (3 %) twice in a row - at most 3 s; still untimed, still outside the W warm-up steps and the K timed ones.
    # (Counter passes of the profiler, which serialise the launches, ask for a short fixed ramp: --prewarm < 100.)
    if args.prewarm >= 100 and not args.dry_run:
        t_end, g, best, calm = time.perf_counter() + 3.0, 50, None, 0
        while time.perf_counter() < t_end and calm < 2:
            t0 = time.perf_counter()
            for _ in range(g):
                h.solve_resident(B)
            h.sync()
            dt = time.perf_counter() - t0
            n_prewarm += g
            if dt < 0.010:
                g = min(5000, int(g * max(2.0, 0.012 / max(dt, 1e-6))))
                continue
            calm = calm + 1 if (best is not None and dt / g > 0.97 * best) else 0
            best = dt / g if best is None else min(best, dt / g)
        barrier()
    for _ in range(args.warmup):
        h.solve_resident(B)
    # The timed region, R times over (a region of K = 20 steps of 0.04 ms is under a millisecond: one perf_counter pair
    # says little, VERDICT r3 "weak" 13).  Each repeat is EXACTLY K steps between a barrier (stream sync + torch sync +
    # collective barrier) and the sync that ends them; the clock stops when this rank's device is idle, the collective
    # barrier that follows is timed on its own (`barrier_ms`) - it is launcher overhead, not solve time (VERDICT r3 "weak" 12).
    dev = "cuda" if (dist is not None and not args.dry_run) else None
    # (at least --repeats, and enough of them for >= 30 ms of timed launches in total: a pilot region sizes it, every rank the same)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.solve_resident(B)
    h.sync()
    pilot = bench_dist.max_over_ranks(dist, time.perf_counter() - t0, device=dev)
    repeats = int(min(400, max(1, args.repeats, -(-0.03 // max(pilot, 1e-6)))))
    dts, mine, barrier_s = [], [], []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            h.solve_resident(B)
        h.sync()
        t1 = time.perf_counter()
        barrier()
        barrier_s.append(time.perf_counter() - t1)
        mine.append(t1 - t0)
        dts.append(bench_dist.max_over_ranks(dist, t1 - t0, device=dev))      # MAX over ranks of every repeat
    med = int(np.argsort(dts)[len(dts) // 2])
    dt = float(dts[med])                                                       # median repeat
    dt_ranks = bench_dist.all_ranks(dist, mine[med], device=dev)
    # The same loop in regions of 200 steps: a region of K = 20 steps pays the ramp and the drain of the
    # launches in flight - the chip holds fewer kernels at its two ends - which a long-running caller does not.  Reported
    # beside `value` (value_200_step_regions, ramp_drain_share), never as it.
    dt200 = None
    if not args.dry_run and args.steps < 200 and not args.no_extra_legs:          # (not under the profiler: 1 000 more launches per pass)
        d2 = []
        for _ in range(5):
            barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                h.solve_resident(B)
            h.sync()
            d2.append(bench_dist.max_over_ranks(dist, time.perf_counter() - t0, device=dev))
        dt200 = float(np.median(d2))

    if args.dry_run:
        # the rehearsal ends here: what the ranks exchange - the gathered result buffer against one process solving the whole
        # batch, the per-rank reports - and the line's multi-rank fields; no figure of it is a measurement
        sol = h.download(B, want_y=False)

        def solve_whole_dry():
            cfg1 = mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX,
                                     scenarios.CAR_LENGTH, circular=True, max_batch=B * world, device=local_rank)
            h1 = emulation.DryHandle(cfg1, settings)
            h1.set_path(tr.kappa, tr.v_ref, tr.ds_next)
            return h1.solve(sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub)
        gather_check = bench_dist.gather_check(dist, rank, sol.u0, sol.status, B * world, solve_whole_dry, device=dev)
        ranks = bench_dist.rank_reports(dist, cfg.device, sol.status, device=dev)
        check_ranks(args, dist, rank, world, ranks)
        if rank == 0:
            out = {"metric": "MPC QP solves/sec (batch, horizon N=%d)" % N, "value": None, "unit": "solves/s", "dry_run": True,
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                   "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "world_size": int(dist.get_world_size()) if dist is not None else 1, "process_group_backend": "gloo" if dist is not None else None,
                   "ms_per_step_by_rank_emulated": [1e3 * t / args.steps for t in dt_ranks], "repeats": repeats,
                   "launches_in_flight": int(args.pipeline), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "config": {"workload": "config%d REHEARSAL: batch=%d per rank, N=%d; CPU emulation of the kernels, gloo" % (args.config, B, N),
                              "batch_per_gpu": B, "horizon": N, "parallelism": "batch-shard x%d" % world},
                   "ranks": ranks, "gather_check": gather_check,
                   "note": "launcher / rank -> device mapping / sharding / barrier / gather rehearsal on a box without GPUs; not a measurement"}
            real_stdout.write(json.dumps(out) + "\n")
            real_stdout.flush()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # per-launch durations of the SAME launch pattern (double-buffered resident launches), HIP events on the streams the
    # kernels are launched on: what rocprofv3 --kernel-trace shows for the timed loop
    n_prof = max(8, min(args.steps, 64))
    each, span = h.solve_resident_profile(B, n_prof)
    # ... and a launch on its own (nothing else on the chip), with the stand-alone assembly kernel beside it
    reps = max(5, min(args.steps, 20))
    ka, ks = [], []
    for _ in range(reps):
        a, s = h.solve_resident_timed(B)
        ka.append(a)
        ks.append(s)
    ms_k1, ms_alone = float(np.mean(ka)), float(np.mean(ks))
    ms_k2 = float(np.mean(each[2:])) if each.size > 4 else float(np.mean(each))      # (the first two start on an empty chip)
    sol = h.download(B, want_y=False)

    # The OTHER operating point (VERDICT r4 "weak" 4): ONE launch in flight - a caller whose step k + 1 needs the result of
    # step k (the reference's own loop, src/simulation.py:134-140) cannot pipeline.  Same K, same repeats, same clock.
    dt1, dts1, same1 = None, [], None
    if not args.no_extra_legs:
        h.set_pipeline(1)
        for _ in range(max(args.warmup, 5)):
            h.solve_resident(B)
        dts1 = []
        for _ in range(repeats):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                h.solve_resident(B)
            h.sync()
            dts1.append(bench_dist.max_over_ranks(dist, time.perf_counter() - t0, device=dev))
        dt1 = float(np.median(dts1))
        sol1 = h.download(B, want_y=False)
        same1 = bool(np.array_equal(sol1.status, sol.status) and np.array_equal(sol1.u0, sol.u0))
        h.set_pipeline(args.pipeline)

    # ... and the algorithm north_star names, run as such: the restated OSQP ADMM iteration at OSQP's own defaults (no polish,
    # no phase 1: mpmpc.stock_settings(), the general kernel) on the same resident batch - its own throughput figure
    admm_stock = None
    if not args.no_extra_legs:
        hs_ = mpmpc.Handle(cfg, mpmpc.stock_settings())
        hs_.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        hs_.set_outputs(want_y=False)
        hs_.upload(wp, x0, cc, lb, ub)
        k_st = max(2, min(args.steps, 8))
        for _ in range(2):
            hs_.solve_resident(B)
        dts_s = []
        for _ in range(3):
            hs_.sync()
            barrier()
            t0 = time.perf_counter()
            for _ in range(k_st):
                hs_.solve_resident(B)
            hs_.sync()
            dts_s.append(bench_dist.max_over_ranks(dist, time.perf_counter() - t0, device=dev))
        sol_s = hs_.download(B, want_y=False)
        hs_.close()
        admm_stock = {"value": world * B * k_st / float(np.median(dts_s)), "unit": "solves/s", "steps": int(k_st), "repeats": 3,
                      "ms_per_step": 1e3 * float(np.median(dts_s)) / k_st,
                      "admm_iters_mean": float(sol_s.iters[:, 0].mean()), "admm_iters_max": int(sol_s.iters[:, 0].max()),
                      "status_counts": {int(a): int(b) for a, b in zip(*np.unique(sol_s.status, return_counts=True))},
                      "kernel": "mpmpc_solve_kernel (general kernel, one instance per wave, Solver::admm)",
                      "note": "the restated OSQP ADMM iteration at OSQP's defaults (eps 1e-3, rho adaptation, no polish, no phase 1: "
                              "mpmpc.stock_settings()) on the same resident batch, launches pipelined like `value`; a launch ends with its "
                              "slowest instance (max-iter instances: 4 000 iterations)"}

    # Two batches in flight (reported beside `value`, never as it): a second handle - its own stream, its own copy of the batch -
    # and the K steps go to the two handles in turn, as a serving loop that double-buffers its batches would issue them.  The
    # launches of the two streams share the chip: where a launch leaves SIMDs or issue slots idle (one wave per SIMD at
    # B = 1 024; the tail launch of configs 4 / 5) the other one's waves run there.
    pipelined = None
    if args.pipelined and B <= 16384:
        h2 = mpmpc.Handle(cfg, settings)
        h2.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        h2.set_packing(args.lanes)
        h2.set_outputs(want_y=False)
        h2.upload(wp, x0, cc, lb, ub)
        pair = (h, h2)
        for i in range(2 * max(args.warmup, 10)):
            pair[i & 1].solve_resident(B)
        h.sync()
        h2.sync()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            pair[i & 1].solve_resident(B)
        h.sync()                 # (both handles: ADVICE r3 - the clock used to stop after the second one's sync only)
        h2.sync()
        barrier()
        dt2 = bench_dist.max_over_ranks(dist, time.perf_counter() - t0, device=dev)
        sol2 = h2.download(B, want_y=False)
        pipelined = {"value": world * B * args.steps / dt2, "unit": "solves/s", "ms_per_step": 1e3 * dt2 / args.steps, "handles": 2,
                     "same_answers": bool(np.array_equal(sol2.status, sol.status) and np.array_equal(sol2.u0, sol.u0)),
                     "note": "the same K steps issued to two handles in turn (two batches of the configuration in flight, each "
                             "launch still one batch): what a double-buffered serving loop gets; `value` above is ONE batch in flight"}
        h2.close()

    # one result buffer, outside the timed region: the shards' controls gathered on every rank (the only data
    # collective a caller of the sharded path may want), checked on rank 0 against ONE process solving the whole
    # world*B batch of the same seed on its own GPU
    def solve_whole():
        cfg1 = mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                                 scenarios.AY_MAX, scenarios.CAR_LENGTH, circular=True, max_batch=B * world,
                                 device=local_rank)
        h1 = mpmpc.Handle(cfg1, settings)
        h1.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        one = h1.solve(sc_all.wp_id, sc_all.x0, sc_all.cc_prev, sc_all.lb, sc_all.ub)
        h1.close()
        return one

    gather_check = bench_dist.gather_check(dist, rank, sol.u0, sol.status, B * world, solve_whole, device=dev)
    ranks = bench_dist.rank_reports(dist, cfg.device, sol.status, device=dev)
    # a multi-rank line is only printed for what --gpus promised: a process group of that many ranks, every
    # rank on a device of its own - otherwise the run ends non-zero WITHOUT a line (a one-GPU figure must not pass for N GPUs)
    check_ranks(args, dist, rank, world, ranks)

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": "MPC QP solves/sec (batch, horizon N=%d)" % N,
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "world_size": int(dist.get_world_size()) if dist is not None else 1,
            "ms_per_step_by_rank": [1e3 * t / args.steps for t in dt_ranks],
            "repeats": repeats, "ms_per_step_min": 1e3 * float(min(dts)) / args.steps, "ms_per_step_max": 1e3 * float(max(dts)) / args.steps,
            "value_from": "median of %d repeats of the timed region of exactly %d steps (max over ranks of every repeat)" % (repeats, args.steps),
            "barrier_ms": 1e3 * float(np.median(barrier_s)),
            "launches_in_flight": int(args.pipeline), "prewarm": int(n_prewarm), "prewarm_fixed": int(args.prewarm),
            "value_200_step_regions": (world * B * 200 / dt200) if dt200 else world * B * args.steps / dt,
            "ramp_drain_share": (1.0 - (world * B * args.steps / dt) / (world * B * 200 / dt200)) if dt200 else 0.0,
            "value_one_launch_in_flight": (world * B * args.steps / dt1) if dt1 else None,
            "one_launch_in_flight": {"value": (world * B * args.steps / dt1) if dt1 else None, "unit": "solves/s", "ms_per_step": (1e3 * dt1 / args.steps) if dt1 else None,
                                     "repeats": repeats, "ms_per_step_min": (1e3 * float(min(dts1)) / args.steps) if dts1 else None,
                                     "ms_per_step_max": (1e3 * float(max(dts1)) / args.steps) if dts1 else None, "same_answers": same1,
                                     "note": "mpmpc_set_pipeline(h, 1): every launch waits for the one before, as in a loop whose step "
                                             "k + 1 needs step k's control (src/simulation.py:134-140); same K, same repeats as `value`"},
            "admm_stock_mode": admm_stock if admm_stock is not None else {"value": None, "note": "skipped: --no-extra-legs"},
            "config": {"workload": "config%d: batch=%d independent poses per GPU, %s weights, N=%d, %s corridor; %s" %
                                   (args.config, B, sc_all.weights, N, "obstacle" if sc_all.obstacles else "free",
                                    algorithm_text(cfg, settings)),
                       "batch_per_gpu": B, "horizon": N, "parallelism": "batch-shard x%d" % world},
        }
        if pipelined:
            out["pipelined_two_handles"] = pipelined
        out["rccl_world_size"] = int(dist.get_world_size()) if dist is not None else 0      # 0: no process group (one process, no launcher)
        out["ranks"] = ranks              # per rank: HIP device ordinal of its handle, status counts of its shard
        if gather_check:
            out["gather_check"] = gather_check
        lib_version = h.lib.mpmpc_version().decode()
        out["library"] = lib_version
        bytes_k2 = algorithmic_bytes_per_solve(N) * B
        nat_tt = native_tt_path(cfg, settings)
        nat = native_path(cfg, settings)
        k2_name = "mpmpc_reduced_t_kernel" if nat_tt else ("mpmpc_reduced_kernel" if nat else "mpmpc_solve_kernel")
        nat = nat or nat_tt
        traffic, traffic_src = pmc_traffic_bytes(k2_name, B, lib_version, args.config)
        # ms_k2: average duration of a launch of the timed pattern (HIP events around each launch on its own stream; with two
        # launches in flight each takes longer than alone and two run side by side); span: first start to last end of the
        # n_prof launches.  `achieved` is the chip's rate over that region - algorithmic bytes of all its launches / span -
        # which with one launch in flight is bytes / avg_ms.
        prof_ms, prof_src = rocprof_kernel_average(args.config, B, lib_version, args.steps)
        chip_rate = bytes_k2 * n_prof / (span * 1e-3)
        out["roofline"] = {"bound": "hbm", "kernel": k2_name + (" (+ tail launches where a launch leaves a tail: mpmpc_reduced_tail_kernel, then mpmpc_solve_kernel on what that leaves)" if nat else ""),
                           "achieved": chip_rate / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": chip_rate / HBM_PEAK,
                           # per kernel: the bytes of ONE launch over the mean duration of one launch of the timed pattern (the
                           # launches in flight overlap, so this is below `frac`, which is the chip's rate over the pattern)
                           "frac_per_kernel": bytes_k2 / (ms_k2 * 1e-3) / HBM_PEAK,
                           "frac_per_kernel_rocprof": (bytes_k2 / (prof_ms * 1e-3) / HBM_PEAK) if prof_ms else None,
                           "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes": bytes_k2, "bytes_per_solve": algorithmic_bytes_per_solve(N),
                           "avg_ms": ms_k2, "launches_in_flight": int(args.pipeline), "launches_measured": int(n_prof), "span_ms": span,
                           "avg_ms_launch_alone": ms_alone, "achieved_launch_alone": bytes_k2 / (ms_alone * 1e-3) / 1e9,
                           "avg_ms_rocprof": prof_ms, "avg_ms_rocprof_source": prof_src,
                           "avg_ms_source": "HIP events around each of %d launches of the timed pattern, on the stream each is launched on "
                                            "(mpmpc_solve_resident_profile)" % n_prof,
                           "note": "achieved = algorithmic bytes per launch x launches / (first start .. last end), HIP events; "
                                   "avg_ms = mean duration of one launch in that pattern: %d launch(es) in flight inside the handle "
                                   "(one launch slot each: stream, output block), so avg_ms is about launches_in_flight x ms_per_step and a "
                                   "launch takes longer than alone (avg_ms_launch_alone).  "
                                   "SURVEY 8(d) bytes: 8(7N+3) in + 8(5N+5)+8 out per solve; the launch writes no multipliers here "
                                   "(mpmpc_set_outputs(0)).  K2 is FP64-VALU / dependency-chain bound, not HBM bound (DESIGN.md "
                                   "section 5): see roofline_fp64" % args.pipeline}
        ms_k2 = span / n_prof          # chip time per launch of the pattern: what the FP64 fractions below are taken over
        ipm = sol.iters[:, 1]
        red = reduced_polish(cfg, settings)
        fa = k2_flops(N, sol.status, ipm, "algorithmic", red, nat and not nat_tt, nat_tt)
        fe = k2_flops(N, sol.status, ipm, "executed", red, nat and not nat_tt, nat_tt)
        out["roofline_fp64"] = {"bound": "fp64-valu", "kernel": k2_name, "peak": FP64_VALU_PEAK / 1e12, "unit": "TFLOP/s",
                                "achieved": fa / (ms_k2 * 1e-3) / 1e12, "frac": fa / (ms_k2 * 1e-3) / FP64_VALU_PEAK,
                                "frac_algorithmic": fa / (ms_k2 * 1e-3) / FP64_VALU_PEAK,
                                "frac_executed": fe / (ms_k2 * 1e-3) / FP64_VALU_PEAK,
                                "flops_per_solve_algorithmic": fa / B, "flops_per_solve_executed": fe / B,
                                "reduced_polish": red,
                                "note": "algorithmic = structure-exploiting count of the implemented recurrence (one step per stage "
                                        "and serial sweep, one level per stage of a cyclic-reduction factorisation; 3x3 chains: "
                                        "factor 180, KKT solve 110 flops per stage); executed = wave "
                                        "instructions x stage-holding lanes; both fitted on the census of the emulated lane "
                                        "code (profiles/census.py, bench.py:_FLOPS)"}
        bytes_k1 = k1_bytes_per_solve(N) * B
        # (PMC traffic of K1 from the same committed summary; the config-2 passes at B = 1 024 / 65 536 and the per-config passes)
        k1_traffic, k1_traffic_src = pmc_traffic_bytes("mpmpc_assemble_kernel", B, lib_version, 2 if args.config == 2 else -args.config)
        k1_rate = bytes_k1 / (ms_k1 * 1e-3)
        out["roofline_assembly"] = {"bound": "hbm", "kernel": "mpmpc_assemble_kernel", "on_solve_path": False,
                                    "achieved": k1_rate / 1e9, "peak": HBM_PEAK / 1e9,
                                    "unit": "GB/s", "frac": min(k1_rate, HBM_ACHIEVABLE) / HBM_PEAK, "avg_ms": ms_k1,
                                    "achievable": HBM_ACHIEVABLE / 1e9, "frac_of_achievable": k1_rate / HBM_ACHIEVABLE,
                                    "above_achievable": bool(k1_rate > HBM_ACHIEVABLE), "frac_unclamped": k1_rate / HBM_PEAK,
                                    "algorithmic_bytes": bytes_k1, "traffic": k1_traffic, "traffic_source": k1_traffic_src,
                                    "note": "the stand-alone K1 (mpmpc_assemble, parity / debug export) timed beside the solve "
                                            "launch; the solve launch assembles its own QP in registers.  `frac` is clamped at the "
                                            "guide's measured-achievable HBM rate (6.29 TB/s = 79 % of the peak): a reading above it "
                                            "(above_achievable) means part of the stores was still in the 256 MB Infinity Cache when "
                                            "the event fired, not that HBM moved that much (VERDICT r4 \"weak\" 8)"}
        st, cnt = np.unique(sol.status, return_counts=True)
        out["status_counts"] = {int(s): int(c) for s, c in zip(st, cnt)}
        # iters[:, 0] is OSQP's iteration counter: 1 = the early attempt alone (general kernels: OSQP's first iterate, one KKT
        # solve, as its start; reduced-native kernels: no OSQP iterate at all, the 1 marks the attempt); only an instance that
        # fell back to the full OSQP run reports more
        adm = np.where(sol.iters[:, 0] > settings.early_polish, sol.iters[:, 0], 0) if settings.polish and settings.early_polish > 0 else sol.iters[:, 0]
        out["iters"] = {"start_steps_per_instance": 0 if (nat or not settings.early_start) else (int(settings.early_polish) if settings.polish else 0),
                        "admm_loop_iterations_mean": float(adm.mean()), "admm_loop_iterations_max": int(adm.max()),
                        "instances_in_admm_fallback": int(np.sum(adm > 0)),
                        "ipm_mean": float(sol.iters[:, 1].mean()), "ipm_max": int(sol.iters[:, 1].max()),
                        "ipm_histogram": {int(k): int(v) for k, v in zip(*np.unique(sol.iters[:, 1], return_counts=True))}}
        # the same step from HOST buffers (mpmpc_solve: upload + assembly and solve in one launch + download): reported, never `value`
        t1 = time.perf_counter()
        for _ in range(5):
            h.solve(wp, x0, cc, lb, ub)
        out["host_buffers"] = {"value": 5 * B / (time.perf_counter() - t1), "unit": "solves/s",
                               "note": "PCIe-inclusive rate of mpmpc_solve on one GPU (pageable numpy buffers)"}
        # the same step through the handle's page-locked staging blocks (mpmpc_staging / mpmpc_solve_staged): inputs written
        # in place every call, outputs read in place - no library-side host copies
        try:
            v = h.staging(B)
            rates = {}
            for key, want_z in (("value", True), ("value_controls_only", False)):
                t1 = time.perf_counter()
                for _ in range(5):
                    v["wp_id"][:] = wp; v["x0"][:] = x0; v["cc_prev"][:] = cc; v["lb"][:] = lb; v["ub"][:] = ub
                    h.solve_staged(B, with_rows=True, want_z=want_z, want_y=False)
                rates[key] = 5 * B / (time.perf_counter() - t1)
            agree = bool(np.array_equal(v["status"], sol.status) and np.array_equal(v["u0"], sol.u0))
            out["host_buffers_staged"] = {"value": rates["value"], "value_controls_only": rates["value_controls_only"], "unit": "solves/s",
                                          "same_answers_as_resident_path": agree,
                                          "note": "PCIe-inclusive: the caller fills the handle's pinned staging block (a numpy copy per "
                                                  "call here) and reads u0 / status / z there; controls_only: z is not copied back"}
        except mpmpc.MpmpcError as e:          # batches above the staging limit have no staging blocks
            out["host_buffers_staged"] = {"value": None, "note": str(e)}
        # ... and a STREAM of such calls, three in flight on this device (multi-purpose-mpc_amd/streamed.py: three handles take
        # the batches in turn through mpmpc_staged_begin / mpmpc_staged_end): upload, launch and download of consecutive
        # batches overlap
        try:
            import streamed
            sb = streamed.StreamedBatches(cfg, settings, depth=3, copy=False)      # results read in place, like the staged call above
            sb.set_path(tr.kappa, tr.v_ref, tr.ds_next)
            rates, agree = {}, True
            for key, want_z in (("value", True), ("value_controls_only", False)):
                for s_ in sb.map([(wp, x0, cc, lb, ub)] * 4, want_z=want_z):        # lay the staging blocks out, warm up
                    pass
                t1 = time.perf_counter()
                n_s = 0
                for s_ in sb.map([(wp, x0, cc, lb, ub)] * 24, want_z=want_z):
                    n_s += 1
                    agree = agree and bool(np.array_equal(s_.status, sol.status) and np.array_equal(s_.u0, sol.u0))
                rates[key] = n_s * B / (time.perf_counter() - t1)
            sb.close()
            out["host_buffers_streamed"] = {"value": rates["value"], "value_controls_only": rates["value_controls_only"], "unit": "solves/s",
                                            "calls_in_flight": 3, "same_answers_as_resident_path": agree,
                                            "note": "PCIe-inclusive: 24 host-buffer batches through StreamedBatches (3 handles on this device "
                                                    "in turn, each call = fill the pinned block, mpmpc_staged_begin, ..., mpmpc_staged_end; the "
                                                    "results are read in the pinned block)"}
        except mpmpc.MpmpcError as e:
            out["host_buffers_streamed"] = {"value": None, "note": str(e)}
        if not args.no_cpu and world == 1:        # (the CPU baseline and the parity legs run on rank 0 at N = 1 only)
            sc_rank = scenarios.Scenario(sc_all.name, N, sc_all.weights, sc_all.obstacles, wp, x0, cc, lb, ub)
            base, ref = cpu_baseline(tr, sc_rank)
            out["cpu_baseline"] = base
            # ... and the arithmetic the reference's own call performs (src/MPC.py:159,183: OSQP at its defaults, no polish,
            # no phase 1) through the same C port, on the same sample
            out["cpu_baseline_stock"], ref_stock = cpu_baseline(tr, sc_rank, seconds=6.0, stock=True)
            ns = ref["status"].size
            both = (ref["status"] == 1) & (sol.status[:ns] == 1)
            # The stock weights put no cost on the steering input, on e_psi and on t (src/simulation.py:101-111): the QP is
            # positive SEMI-definite, and where a corridor bound is weakly active its optimum is a face, not a point.  An
            # instance whose control differs by more than the tolerance while both solvers hold a certificate and the
            # objectives agree to 1e-9 relative is an alternative optimum: counted, not compared.
            du = np.abs(sol.u0[:ns] - ref["u0"]).max(axis=1)
            far = np.flatnonzero(both & (du > 1e-6))
            alt = np.zeros(ns, bool)
            if far.size:
                qf = h.assemble(wp[far], x0[far], cc[far], lb[far], ub[far])
                od, orf = qp_objective(qf, N, sol.z[:ns][far]), qp_objective(qf, N, ref["z"][far])
                alt[far[np.abs(od - orf) <= 1e-9 * np.maximum(1.0, np.abs(orf))]] = True
            out["alternative_optima"] = {"count": int(alt.sum()), "max_abs_u_diff": float(du[alt].max()) if alt.any() else 0.0,
                                         "note": "both certified (KKT 1e-8), objectives equal to 1e-9 relative, controls differ by "
                                                 "more than 1e-6: the QP is only positive semi-definite (no cost on the steering "
                                                 "input); excluded from max_abs_u_minus_uref / max_abs_plan_minus_ref"}
            both = both & ~alt
            out["max_abs_u_minus_uref"] = float(np.max(du[both])) if both.any() else None
            # Verdicts, both pairings (VERDICT r3 "weak" 5).  The device's DEFAULT returns a usable plan (status 2) for an instance
            # infeasible by less than OSQP's own primal tolerance, like the reference's solver call does; the certified C port
            # reports every proven infeasibility.  `status_agreement` keeps comparing those two semantics (it is < 1 exactly by
            # the marginal instances); the like-for-like pairings are
            #   strict:  device with phase1_accept = 0  vs  the certified port          (same semantics: must be 1.0)
            #   default: device default, usable (status 1 / 2) or refused  vs  the port run as stock OSQP (status > 0 or not)
            out["status_agreement"] = float(np.mean(ref["status"] == sol.status[:ns]))
            h.set_settings(mpmpc.default_settings(**dict(overrides, phase1_accept=0)))
            strict = h.solve(wp[:ns], x0[:ns], cc[:ns], lb[:ns], ub[:ns])
            h.set_settings(settings)
            n2 = min(ns, ref_stock["status"].size)
            out["status_agreement_pairings"] = {
                "strict_device_vs_certified_port": float(np.mean(strict.status == ref["status"])),
                "default_device_usable_vs_stock_osqp_port_usable": float(np.mean(((sol.status[:n2] == 1) | (sol.status[:n2] == 2)) == (ref_stock["status"][:n2] > 0))),
                # ... and as a COUNT: the instances on the other branch, of how many
                "default_device_vs_stock_osqp_port_disagreeing_instances": int(np.sum(((sol.status[:n2] == 1) | (sol.status[:n2] == 2)) != (ref_stock["status"][:n2] > 0))),
                "default_device_vs_stock_osqp_port_disagreeing_of": int(n2),
                "marginal_instances_status_2": int(np.sum(sol.status[:ns] == 2)), "sample": int(ns), "sample_stock": int(n2)}
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
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
