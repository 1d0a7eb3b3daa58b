"""The kernels of the "next" rows of SURVEY section 8f in one run, for rocprofv3 --kernel-trace --stats:
K0 corridor tables from the map (obstacle grid, 200 start waypoints x 50 columns), K3 closed-loop rollout
(1024 cars, 50 steps: localise, solve, advance), K4 speed profile (Sim_Track, 1 and 1024 paths).

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 profiles/next_rows.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-purpose-mpc_amd"), os.path.join(ROOT, "tests"), ROOT]
import mpmpc            # noqa: E402
import mpmpc_testlib as T   # noqa: E402
import scenarios        # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
g1 = np.load(G + "/g1_path_sim_track.npz")
g3 = np.load(G + "/g3_corridor.npz")
tr = scenarios.sim_track()
hh, ww = g1["grid_shape"]
grid = np.ascontiguousarray(np.unpackbits(g1["grid_obstacles"])[:hh * ww].reshape(hh, ww).astype(np.int8))
sm = float(g3["safety_margin"][0])
B = 1024

h = mpmpc.Handle(T.stock_config(30, max_batch=B))
h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
h.set_map(grid, (-1.0, -2.0), 0.005)
h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
t = time.perf_counter()
for _ in range(20):
    ub, lb, bad = h.build_corridor(50, 2 * sm, sm, want_tables=False)
print("K0: corridor table [200 x 50] from the obstacle map, %.3f ms per rebuild (host call)" % ((time.perf_counter() - t) / 20 * 1e3))

grid_free = np.ascontiguousarray(np.unpackbits(g1["grid_free"])[:hh * ww].reshape(hh, ww).astype(np.int8))
h.set_map(grid_free, (-1.0, -2.0), 0.005)          # the rollout below drives on the free track
h.build_corridor(50, 2 * sm, sm, want_tables=False)
cum = np.cumsum(g1["segment_lengths"])
starts = np.random.default_rng(7).integers(0, 200, B)
poses = np.stack([g1["x"][starts], g1["y"][starts], g1["psi"][starts]], axis=1)
h.rollout_init(0.05, cum, cum[starts], poses)
h.rollout_step(5)
h.sync()
t = time.perf_counter()
h.rollout_step(50)
h.sync()
print("K3: closed loop, %d cars, corridor table built on the device from the free map, %.3f ms per step" % (B, (time.perf_counter() - t) / 50 * 1e3))
h.close()

n = tr.kappa.size - 1
li, kap = np.ascontiguousarray(tr.ds_next[:n]), np.ascontiguousarray(tr.kappa[:n])
lim = np.array([-0.1, 0.5, 0.0, 1.0, 4.0])
for P in (1, 1024):
    L, K, M_ = np.tile(li, (P, 1)), np.tile(kap, (P, 1)), np.tile(lim, (P, 1))
    mpmpc.speed_profile(L, K, M_)
    t = time.perf_counter()
    for _ in range(5):
        v, st, it = mpmpc.speed_profile(L, K, M_)
    print("K4: speed profile, %d path(s) of %d speeds: %.3f ms per call, status %s" % (P, n, (time.perf_counter() - t) / 5 * 1e3, np.unique(st)))
