"""Host side of the pipelined resident launches: time to ENQUEUE K launches (no sync) and until they are done, by pipeline depth.
    /usr/local/graft/bin/gpurun -- 'python profiles/launch_rate.py'"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-purpose-mpc_amd"), ROOT]
import mpmpc, scenarios

tr = scenarios.sim_track()
for B in (1024, 256):
    sc = scenarios.make(2, tr, B=B)
    Q, R, QN = scenarios.WEIGHTS[sc.weights]
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=B)
    h = mpmpc.Handle(cfg, mpmpc.default_settings())
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_outputs(False)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for depth in (1, 2, 3, 4, 6, 8):
        h.set_pipeline(depth)
        for _ in range(300):
            h.solve_resident(B)
        h.sync()
        K = 400
        t0 = time.perf_counter()
        for _ in range(K):
            h.solve_resident(B)
        t1 = time.perf_counter()
        h.sync()
        t2 = time.perf_counter()
        print("B %5d depth %d: enqueue %.2f us per launch, done %.2f us per launch -> %.1f M solves/s" % (B, depth, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, B * K / (t2 - t0) / 1e6))
    h.close()
