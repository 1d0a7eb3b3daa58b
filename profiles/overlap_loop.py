"""Two handles fed in turn, nothing else: the loop profiles/overlap_trace.sh puts under rocprofv3 --kernel-trace.
    python profiles/overlap_loop.py CONFIG STEPS"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-purpose-mpc_amd"), ROOT]
import mpmpc        # noqa: E402
import scenarios    # noqa: E402

conf, steps = int(sys.argv[1]), int(sys.argv[2])
tr = scenarios.sim_track()
sc = scenarios.make(conf, tr, B=scenarios.CONFIGS[conf].get("B_per_gpu"))      # (config 5: one GPU's shard of the sweep)
B = sc.B
Q, R, QN = scenarios.WEIGHTS[sc.weights]
cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=B)
hs = [mpmpc.Handle(cfg) for _ in range(2)]
for h in hs:
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_outputs(False)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(50):
        h.solve_resident(B)
    h.sync()
t0 = time.perf_counter()
for i in range(steps):
    hs[i & 1].solve_resident(B)
for h in hs:
    h.sync()
dt = time.perf_counter() - t0
print("config %d B %d: %d steps on two handles in turn, %.2f M solves/s by the host clock" % (conf, B, steps, steps * B / dt / 1e6))
