"""Which branch of src/MPC.py:185-216 does get_control take - a fresh plan or the fallback - with the DEFAULT settings of the
device, against the restated stock OSQP (the C oracle at OSQP's defaults: the arithmetic of the reference's own solver call,
src/MPC.py:159,183)?  Lists every instance of configs 4 / 5 on which the two disagree: phase 1's least violation (resid[0] of
the device), OSQP's status / iteration count / primal residual.  VERDICT r4 item 3.

    python profiles/branch_agreement.py [config ...] > profiles/r5/branch_agreement.txt        (on a GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "multi-purpose-mpc_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

import mpmpc  # noqa: E402
import mpmpc_testlib as T  # noqa: E402
import oracle_c as OC  # noqa: E402
import scenarios  # noqa: E402


def compare(cfgid, B, settings=None):
    """-> dict(agreement, rows): rows = (instance, device status, device violation, stock status, stock iterations, stock pri_res)"""
    track = scenarios.sim_track()
    sc = scenarios.make(cfgid, track, B=B)
    cfg = T.stock_config(sc.N, sc.weights, max_batch=B)
    st = settings or mpmpc.default_settings()
    h = mpmpc.Handle(cfg, st)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    h.close()
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN, scenarios.XMAX, 4.0, 0.12)
    stock = OC.mpc_batch(ocfg, OC.settings(polish=0, early_polish=0, phase1=0), track.kappa, track.v_ref, track.ds_next, sc.wp_id,
                         sc.x0, sc.cc_prev, sc.lb, sc.ub)
    usable = lambda s: np.isin(s, (1, 2, -2))
    dis = np.flatnonzero(usable(sol.status) != usable(stock["status"]))
    rows = [(int(i), int(sol.status[i]), float(sol.resid[i, 0]), int(stock["status"][i]), int(stock["iters"][i, 0]), float(stock["resid"][i, 0]))
            for i in dis]
    return dict(agreement=1.0 - dis.size / B, rows=rows, device=dict(zip(*map(lambda a: a.tolist(), np.unique(sol.status, return_counts=True)))),
                stock=dict(zip(*map(lambda a: a.tolist(), np.unique(stock["status"], return_counts=True)))), B=B,
                threshold=1e-3 + 1e-3 * float(scenarios.UMAX[1]))


if __name__ == "__main__":
    print("# %s" % mpmpc.load_library().mpmpc_version().decode())
    for c in [int(a) for a in sys.argv[1:]] or [4, 5]:
        for B in ((8192,) if c == 4 else (8192, 65536)):
            r = compare(c, B)
            print("config %d, B = %d: device %s, restated stock OSQP %s" % (c, B, r["device"], r["stock"]))
            print("  same branch on %d of %d instances (%.5f); OSQP's primal tolerance at the steering limit: %.4e" % (B - len(r["rows"]), B, r["agreement"], r["threshold"]))
            print("  instance  device  least violation   stock  iterations  pri_res      why")
            for i, ds, dv, ss, si, sp in r["rows"]:
                why = "OSQP abandoned at max_iter: neither of its tests passed, it returns its iterate" if si >= 4000 else \
                      ("within %.2f %% of the tolerance: OSQP stopped an iterate short of its limit" % (100 * abs(dv / r["threshold"] - 1)) if abs(dv / r["threshold"] - 1) < 0.05 else "?")
                print("  %8d  %6d  %.4e        %5d  %10d  %.4e   %s" % (i, ds, dv, ss, si, sp, why))
