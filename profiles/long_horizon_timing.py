"""Launch latency and throughput of the solve kernels around the 63 / 64 horizon boundary (one wavefront vs a workgroup of
2 / 4 wavefronts per instance), stock weights, free corridor:  python profiles/long_horizon_timing.py > profiles/r5/long_horizon.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "multi-purpose-mpc_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

import mpmpc  # noqa: E402
import mpmpc_testlib as T  # noqa: E402
import scenarios  # noqa: E402

emu = T.Emul()
track = scenarios.sim_track()
print("# %s" % mpmpc.load_library().mpmpc_version().decode())
print("# N, kernel, B: ms per launch (resident, one launch in flight), solves/s; interior-point iterations mean")
# (lanes: 0 = the launcher's choice - at horizons 64 .. 127 TWO stages per lane in one wavefront, at 128 .. 255 on a workgroup of two;
#  128 / 256 = round 5's one-stage workgroup kernels)
for N, native, lanes in ((50, 1, 0), (63, 1, 0), (63, 0, 0), (64, 1, 0), (64, 1, 128), (100, 1, 0), (100, 1, 128), (127, 1, 0), (127, 1, 128), (128, 1, 0), (128, 1, 256), (200, 1, 0), (200, 1, 256), (255, 1, 0), (255, 1, 256)):
    tw = T.wide_track(track, emu, max(N, 50))
    for B in (64, 1024, 8192):
        sc = scenarios.make(2, tw, B=B, N=N)
        cfg = T.stock_config(N, sc.weights, max_batch=B)
        h = mpmpc.Handle(cfg, mpmpc.default_settings(native=native))
        h.set_path(track.kappa, track.v_ref, track.ds_next)
        h.set_outputs(False)
        if lanes:
            h.set_packing(lanes)
        h.set_pipeline(1)
        h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        for _ in range(3):
            h.solve_resident(B)
        h.sync()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            h.solve_resident(B)
        h.sync()
        dt = (time.perf_counter() - t0) / n
        sol = h.download(B)
        h.close()
        kern = ("two stages per lane, one wavefront" if 63 < N < 128 and not lanes else ("two stages per lane, workgroup of 128" if N > 127 and not lanes else "workgroup of %d lanes" % mpmpc.stage_ld(N))) if N > 63 else ("reduced-native wavefront kernel" if native else "general wavefront kernel")
        print("N %3d  %-32s B %5d: %8.3f ms  %10.0f solves/s  ipm %.2f  solved %d" % (N, kern, B, dt * 1e3, B / dt, sol.iters[:, 1].mean(), int((sol.status == 1).sum())))
