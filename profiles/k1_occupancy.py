"""K1 at 4 / 5 / 6 / 8 waves per SIMD (builds -DMPMPC_K1_WAVES=w in profiles/_ab/K1w<w>.so), same box, K1 launched alone back to back
(mpmpc_solve_resident_timed interleaves it with K2, whose dirty output lines K1's writes then have to push out of the cache) and
interleaved:   /usr/local/graft/bin/gpurun --timeout 600 -- 'python profiles/k1_occupancy.py'"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path[:0] = [os.path.join(%r, "multi-purpose-mpc_amd"), os.path.join(%r, "tests"), %r]
import numpy as np, mpmpc, scenarios
w = sys.argv[1]
mpmpc._lib = mpmpc.load_library(os.path.join(%r, "profiles", "_ab", "K1w%%s.so" %% w))
tr = scenarios.sim_track()
for B in (8192, 65536):
    sc = scenarios.make(2, tr, B=B)
    Q, R, QN = scenarios.WEIGHTS[sc.weights]
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=B)
    h = mpmpc.Handle(cfg, mpmpc.default_settings())
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next); h.set_outputs(False)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(50): h.solve_resident(B)
    h.sync()
    ks = [h.solve_resident_timed(B)[0] for _ in range(30)]
    byts = (8 * (7 * sc.N + 3) + 8 * 27 * (sc.N + 1)) * B
    alone = h.assemble_timed(B, 30) if hasattr(h, "assemble_timed") else [float("nan")]
    print("waves %%s B %%6d: K1 after K2 min %%.4f med %%.4f ms -> %%.3f / %%.3f of 8 TB/s | alone, back to back min %%.4f med %%.4f -> %%.3f / %%.3f" %% (
        w, B, min(ks), np.median(ks), byts / min(ks) / 1e-3 / 8e12, byts / np.median(ks) / 1e-3 / 8e12,
        min(alone), np.median(alone), byts / min(alone) / 1e-3 / 8e12, byts / np.median(alone) / 1e-3 / 8e12))
    h.close()
''' % (ROOT, ROOT, ROOT, ROOT)
for rep in range(2):
    for w in sys.argv[1:] or ("4", "5", "6", "8"):
        r = subprocess.run([sys.executable, "-c", CHILD, w], capture_output=True, text=True)
        sys.stdout.write(r.stdout + (r.stderr[-400:] if r.returncode else ""))
        sys.stdout.flush()
