"""K1 (stand-alone assembly kernel) at HBM-resident sizes, grid cap and store flavour A/B ON ONE BOX (boxes differ by up to
1.35 x in what this kernel reaches: compare within a run only):
    /usr/local/graft/bin/gpurun --timeout 600 -- 'python profiles/k1_timing.py'
Each combination runs in its own process (the knobs are read once per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path[:0] = [os.path.join(%r, "multi-purpose-mpc_amd"), %r]
import numpy as np, mpmpc, scenarios
tr = scenarios.sim_track()
for B in (8192, 65536):
    sc = scenarios.make(2, tr, B=B)
    Q, R, QN = scenarios.WEIGHTS[sc.weights]
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=B)
    h = mpmpc.Handle(cfg, mpmpc.default_settings())
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next); h.set_outputs(False)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(100): h.solve_resident(B)
    h.sync()
    ks = [h.solve_resident_timed(B)[0] for _ in range(30)]
    byts = (8 * (7 * sc.N + 3) + 8 * 27 * (sc.N + 1)) * B
    print("blocks %%s nt %%s B %%6d: K1 min %%.4f med %%.4f ms -> %%.3f / %%.3f of 8 TB/s" %% (os.environ.get("MPMPC_K1_BLOCKS"), os.environ.get("MPMPC_K1_NT"), B, min(ks), np.median(ks), byts / min(ks) / 1e-3 / 8e12, byts / np.median(ks) / 1e-3 / 8e12))
    h.close()
''' % (ROOT, ROOT)
# (the grid-size knob MPMPC_K1_BLOCKS of the first version of this script is gone with the kernel's grid-stride loop: see
#  profiles/k1_occupancy.py and profiles/r4/k1_occupancy.txt for what replaced that experiment)
for rep in range(2):
    for nt in ("0", "1"):
        env = dict(os.environ, MPMPC_K1_NT=nt)
        sys.stdout.write(subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout)
        sys.stdout.flush()
