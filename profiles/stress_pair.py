"""Randomised device-vs-emulation sweep of the layouts with TWO stages per lane (round 6; run on the GPU box):
    python profiles/stress_pair.py [seed]
 * horizons 16 .. 31 with mpmpc_set_packing(h, 16) - four instances per wavefront, ragged batches, both verdict semantics;
 * horizons 64 .. 255 with the launcher's own choice (one wavefront per instance up to 127, a workgroup of two above; the tail
   kernels behind them), stock and - up to 127 - time-optimal weights.
Every combination must agree with the emulation of the same lane code in status and (to the device's reciprocal seeds: one or two
interior-point iterations) iteration counts, in z / u0 to 1e-9, and carry its own certificates (KKT for every 1, Farkas for every -3)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("multi-purpose-mpc_amd", "tests", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
import mpmpc            # noqa: E402
import mpmpc_testlib as T   # noqa: E402
import scenarios        # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tr = scenarios.sim_track()
em = T.Emul()
worst, n_inst, bad, notes, trials = 0.0, 0, 0, 0, 0
plan = [("pair16", int(rng.integers(16, 32)), int(rng.choice([2, 4]))) for _ in range(24)] + \
       [("long", int(rng.integers(64, 256)), int(rng.choice([2, 4]))) for _ in range(14)] + \
       [("long_tt", int(rng.integers(64, 128)), 3) for _ in range(6)]
for kind, N, cfg_id in plan:
    trials += 1
    B = int(rng.integers(1, 200 if kind == "pair16" else 40))
    weights = scenarios.CONFIGS[cfg_id]["weights"]
    tw = T.wide_track(tr, em, N) if N > 50 else tr
    sc = scenarios.make(cfg_id, tw, B=B, N=N)
    perm = rng.permutation(B)
    wp, x0, cc, lb, ub = sc.wp_id[perm], sc.x0[perm], sc.cc_prev[perm], sc.lb[perm], sc.ub[perm]
    x0 = x0 + rng.normal(0, 0.01, x0.shape) * np.array([1.0, 1.0, 0.0])
    cfg = T.stock_config(N, weights, max_batch=B)
    st = mpmpc.default_settings(phase1_accept=trials % 2)
    h = mpmpc.Handle(cfg, st)
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    qp = em.assemble(cfg, tw, (wp, x0, cc, lb, ub))
    if kind == "pair16":
        h.set_packing(16)
        emu, _ = em.solve_launch(cfg, st, qp, G=16)
    else:
        emu = em.solve(cfg, st, qp)                 # the launcher's sequence of kernels at this horizon
    dev = h.solve(wp, x0, cc, lb, ub, want_y=True)
    h.close()
    it_ok = (dev.iters == emu.iters).all(axis=1) | ((dev.status == -3) & (np.abs(dev.iters - emu.iters).max(axis=1) <= 1)) | \
            ((dev.status == emu.status) & (dev.iters[:, 0] == emu.iters[:, 0]) & (np.abs(dev.iters[:, 1] - emu.iters[:, 1]) <= 2))
    same = np.array_equal(dev.status, emu.status) and bool(it_ok.all())
    notes += int((~(dev.iters == emu.iters).all(axis=1)).sum())
    ok = dev.status == 1
    dz = float(np.abs(dev.z[ok] - emu.z[ok]).max()) if ok.any() else 0.0
    du = float(np.abs(dev.u0[ok] - emu.u0[ok]).max()) if ok.any() else 0.0
    worst = max(worst, dz, du)
    cert = True
    if ok.any():
        prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, dev.z[ok], dev.y[ok])
        cert = cert and max(prim.max(), stat.max(), comp.max()) <= 1e-8
    inf = dev.status == -3
    if inf.any():
        cert = cert and bool(T.farkas_batch(qp[:, inf, :], N, dev.y[inf])[0].all())
    cert = cert and set(np.unique(dev.status)) <= ({1, -3} if st.phase1_accept == 0 else {1, 2, -3})
    tol = 1e-9 if kind != "long_tt" else 1e-8
    if not same or dz > tol or du > tol or not cert:
        bad += 1
        print("MISMATCH %s N=%d cfg=%d B=%d: status/iters equal %s, dz %.2e du %.2e, certificates %s" % (kind, N, cfg_id, B, same, dz, du, cert))
        d = np.flatnonzero((dev.status != emu.status) | (dev.iters != emu.iters).any(axis=1))
        for i in d[:4]:
            print("   instance %d: device status %d iters %s | emulation status %d iters %s" % (i, dev.status[i], dev.iters[i], emu.status[i], emu.iters[i]))
    else:
        print("ok %-8s N=%3d cfg=%d B=%3d  statuses %s" % (kind, N, cfg_id, B, dict(zip(*map(lambda a: a.tolist(), np.unique(dev.status, return_counts=True))))))
    n_inst += B
print("trials %d, instances %d, mismatches %d, worst |device - emulation| %.2e, interior-point iteration counts apart (same status, same point): %d" % (trials, n_inst, bad, worst, notes))
sys.exit(1 if bad else 0)
