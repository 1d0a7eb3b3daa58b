"""Two stages per lane on the device (mpmpc_reduced_pair_kernel, set_packing(16) at 17 .. 32 stages) against the shipped one-stage
layout: same statuses / iteration counts / controls, KKT on its own output, and the rates of both on one box.
    python profiles/s2_check.py [--quick]"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "multi-purpose-mpc_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

import mpmpc  # noqa: E402
import mpmpc_testlib as T  # noqa: E402
import scenarios  # noqa: E402

track = scenarios.sim_track()
print("# %s" % mpmpc.load_library().mpmpc_version().decode())


def run(cfgid, B, lanes, depth, steps=50, N=None, check=None):
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = T.stock_config(sc.N, sc.weights, max_batch=B)
    h = mpmpc.Handle(cfg, mpmpc.default_settings())
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    h.set_packing(lanes)
    h.set_outputs(False)
    h.set_pipeline(depth)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(200):
        h.solve_resident(B)
    h.sync()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(steps):
            h.solve_resident(B)
        h.sync()
        ts.append((time.perf_counter() - t0) / steps)
    sol = h.download(B)
    if check is not None:
        inp = (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        qp = h.assemble(*inp)
        full = h.solve(*inp, want_y=True)
        ok = full.status == 1
        prim, stat, comp = T.kkt_batch(qp[:, ok, :], sc.N, full.z[ok], full.y[ok])
        okb = (check.status == 1) & ok
        print("    lanes %d vs %d: statuses equal %s, iterations equal %s, max|u0 - u0'| %.2e, KKT %.1e %.1e %.1e, resident = host-buffer call %s"
              % (lanes, 32, np.array_equal(full.status, check.status), np.array_equal(full.iters, check.iters),
                 np.abs(full.u0[okb] - check.u0[okb]).max(), prim.max(), stat.max(), comp.max(),
                 np.array_equal(sol.status, full.status) and np.array_equal(sol.u0, full.u0)))
    else:
        inp = (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        sol = h.solve(*inp, want_y=True)
    h.close()
    dt = float(np.median(ts))
    print("config %d N %d B %6d lanes %2d depth %d: %8.4f ms/step  %7.2f M solves/s  ipm %.2f/%d" % (cfgid, sc.N, B, lanes, depth, dt * 1e3, B / dt / 1e6,
          sol.iters[:, 1].mean(), sol.iters[:, 1].max()))
    return sol


quick = "--quick" in sys.argv
for cfgid, B in ((2, 1024), (4, 8192), (2, 65536)) if not quick else ((2, 1024),):
    steps = 200 if B <= 1024 else (30 if B <= 8192 else 8)
    for depth in (4, 1):
        ref = run(cfgid, B, 32, depth, steps)
        run(cfgid, B, 16, depth, steps, check=ref if depth == 4 else None)
if not quick:
    for N in (16, 20, 25, 31):
        ref = run(4, 2048, 32, 4, 30, N=N)
        run(4, 2048, 16, 4, 30, N=N, check=ref)
