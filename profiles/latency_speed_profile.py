"""Latency of mpmpc_speed_profile (K4) for 1 .. 16384 copies of the Sim_Track path:  python profiles/latency_speed_profile.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-purpose-mpc_amd"), os.path.join(ROOT, "tests"), ROOT]
import mpmpc, scenarios
tr = scenarios.sim_track()
n = tr.kappa.size - 1
li = np.ascontiguousarray(tr.ds_next[:n])
kap = np.ascontiguousarray(tr.kappa[:n])
lim = np.array([-0.1, 0.5, 0.0, 1.0, 4.0])
for B in (1, 64, 1024, 16384):
    L = np.tile(li, (B, 1)); K = np.tile(kap, (B, 1)); M = np.tile(lim, (B, 1))
    mpmpc.speed_profile(L, K, M)
    t = time.perf_counter()
    for _ in range(5):
        v, st, it = mpmpc.speed_profile(L, K, M)
    dt = (time.perf_counter() - t) / 5
    print("B=%6d: %.3f ms per call, %.1f us per path, status %s iters %s" % (B, dt * 1e3, dt * 1e6 / B, np.unique(st), it[:1]))
