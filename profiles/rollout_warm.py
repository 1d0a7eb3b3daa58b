"""Closed-loop rollout on the device with and without the warm start (previous step's shifted active set):
per-step time for B cars and how far the trajectories drift apart.  python profiles/rollout_warm.py  (GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("multi-purpose-mpc_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import mpc_np as M          # noqa: E402
import mpmpc                # noqa: E402
import mpmpc_testlib as T   # noqa: E402
import scenarios            # noqa: E402

g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
g3 = np.load(M.GOLDEN + "/g3_corridor.npz")
N, steps = 30, 60
tr = scenarios.sim_track()
cum = np.cumsum(g1["segment_lengths"])
for B in (8, 1024, 8192):
    rng = np.random.default_rng(7)
    starts = rng.integers(0, 200, B)
    res = {}
    for warm in (False, True, "auto"):
        h = mpmpc.Handle(T.stock_config(N, max_batch=B))
        h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        h.set_corridor(g3["ub_free"], g3["lb_free"])
        h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
        poses = np.stack([g1["x"][starts], g1["y"][starts], g1["psi"][starts]], axis=1)
        h.rollout_warm_start(warm)
        h.rollout_init(0.05, cum, cum[starts], poses)
        h.rollout_step(5)
        h.sync()
        t = time.perf_counter()
        h.rollout_step(steps)
        h.sync()
        dt = (time.perf_counter() - t) / steps
        st_ = h.rollout_state()
        sol_ = h.download(B)
        st_["warm_hits"] = float(np.mean(sol_.iters[:, 0] == 0))
        res[warm] = (dt, st_)
        h.close()
    print("         default (auto): %.3f ms/step" % (res["auto"][0] * 1e3))
    a, b = res[False][1], res[True][1]
    print("B=%5d: %.3f ms/step cold, %.3f ms/step warm (x%.1f); after %d steps max |ds| %.2e, max |dpose| %.2e, alive %d / %d, counters %d / %d" %
          (B, res[False][0] * 1e3, res[True][0] * 1e3, res[False][0] / res[True][0], steps + 5, np.max(np.abs(a["s"] - b["s"])),
           np.max(np.abs(a["pose"] - b["pose"])), (a["alive"] == 1).sum(), (b["alive"] == 1).sum(), a["counter"].sum(), b["counter"].sum()))
    print("         share of cars certified from the warm start in the last step: %.3f" % b["warm_hits"])
    print("         statuses of the last step, cold:", dict(zip(*np.unique(a["status"], return_counts=True))), " warm:", dict(zip(*np.unique(b["status"], return_counts=True))))
