"""G8 - the INDEPENDENT solve goldens: for the QPs of the G4 captures (what the REFERENCE handed to
osqp.setup, tests/golden/g4_assembly_N*.npz) and for the first 128 instances of BASELINE configs 2, 3, 4, the point the
independent leg of the oracle reaches - oracle/independent.py: restated OSQP ADMM to eps = 1e-10, then ONE stock OSQP
polish; no interior point, no step indicators, no active-set rounds, no phase 1: nothing of the device's algorithm -
with its own KKT residuals, and what scipy's bundled HiGHS returns for the same QP (point and objective).

    python tests/golden/make_g8.py [workers] [names]  (system python; needs no reference and no GPU; ~10 min on 8 cores;
                                                       names: comma-separated subset, e.g. cfg3,g4_N50 - round 4 regenerated
                                                       exactly these two, with the primal active-set method, see one_active_set)

Writes tests/golden/g8_independent_{g4_N3,g4_N10,g4_N30,g4_N50,cfg2,cfg3,cfg4}.npz.  The tests compare the DEVICE with
these files (tests/test_gpu_parity.py, tests/test_emul_parity.py); nothing here may be touched by a device commit."""
import multiprocessing as mp
import os
import sys

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), ROOT]
import independent as I  # noqa: E402
import mpc_np as M       # noqa: E402


def one_active_set(args):
    """N = 50 (round 4): the restated ADMM does not reach 1e-10 within 10^6 iterations on most of these
    QPs, so ONE polish of its iterate certified only 82 / 128 (cfg3) and 32 / 64 (g4_N50).  These two files come from the
    independent leg's textbook primal active-set method instead (oracle/independent.py:solve_primal_active_set), started at the
    feasible point nearest to a cheap ADMM iterate (eps = 1e-5, at most 20 000 iterations)."""
    import oracle_c as OC
    Pd, q, A, l, u = args
    st = OC.settings(polish=0, early_polish=0, phase1=0, eps_abs=1e-5, eps_rel=1e-5, max_iter=20000)
    xa, ya, info = OC.solve(np.diag(Pd), q, A, l, u, st)
    r = I.solve_primal_active_set(Pd, q, A, l, u, x_near=xa if np.all(np.isfinite(xa)) else None)
    ok = r["status"] == 1 and max(r["kkt"]) <= 1e-8
    return (r["x"], r["status"], int(info.iters), int(ok), np.array(r["kkt"]), np.full(q.size, np.nan), np.nan)


def one(args):
    Pd, q, A, l, u = args
    r = I.solve_admm_polish(Pd, q, A, l, u, max_iter=1000000)
    # (HiGHS' QP solver loops on some N = 50 problems, printing "dimension mismatch" until its time limit: horizons up to
    #  30 only, like golden G5)
    hx, ho = I.highs_solution(Pd, q, A, l, u, time_limit=5.0) if q.size <= 5 * 30 + 3 else (None, np.nan)
    return (r["x"], r["status"], r["admm_iters"], r["polished"], np.array(r["kkt"]), hx if hx is not None else np.full(q.size, np.nan), ho)


ACTIVE_SET = ("g4_N50", "cfg3")          # the files produced by the primal active-set method
ONLY = None                              # names to (re)generate; None = all


def run(name, qps, pool, meta):
    if ONLY is not None and name not in ONLY:
        return
    res = pool.map(one_active_set if name in ACTIVE_SET else one, qps, chunksize=1)
    X = np.array([r[0] for r in res])
    st = np.array([r[1] for r in res], np.int32)
    out = dict(x=X, status=st, admm_iters=np.array([r[2] for r in res], np.int32), polished=np.array([r[3] for r in res], np.int32),
               kkt=np.array([r[4] for r in res]), x_highs=np.array([r[5] for r in res]), obj_highs=np.array([r[6] for r in res]), **meta,
               note=np.array(["oracle/independent.py: textbook primal active-set method (solve_primal_active_set) from the feasible point nearest "
                              "to a restated-OSQP iterate at 1e-5; polished = 1: its point passed the KKT test at 1e-8; admm_iters: iterations of "
                              "that start; no HiGHS at this horizon" if name in ACTIVE_SET else
                              "oracle/independent.py: restated OSQP ADMM to 1e-10 (max 1e6 iterations) + ONE stock polish; polished = 1: the "
                              "polished point passed the KKT test at 1e-8 and is stored, else the ADMM iterate is; HiGHS: scipy's bundled QP solver"]))
    np.savez_compressed(os.path.join(HERE, "g8_independent_%s.npz" % name), **out)
    ok = st > 0
    print("%-8s %3d QPs: ADMM verdicts %s, polished %d, worst KKT of the polished %.1e, ADMM iterations median %d / max %d, HiGHS answers %d" %
          (name, len(qps), dict(zip(*np.unique(st, return_counts=True))), int(out["polished"].sum()),
           float(np.nanmax(np.where(out["polished"][:, None] == 1, out["kkt"], 0.0))), int(np.median(out["admm_iters"])), int(out["admm_iters"].max()),
           int(np.isfinite(out["obj_highs"]).sum())), flush=True)
    del ok


def main():
    global ONLY
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if len(sys.argv) > 2:
        ONLY = tuple(sys.argv[2].split(","))
    with mp.Pool(workers) as pool:
        for N in (3, 10, 30, 50):
            g = np.load(os.path.join(HERE, "g4_assembly_N%d.npz" % N))
            n, m = 5 * N + 3, 8 * N + 6
            qps = []
            for c in range(g["s"].size):
                lo, hi = g["A_case_ptr"][c], g["A_case_ptr"][c + 1]
                A = sparse.csc_matrix((g["A_data"][lo:hi], g["A_indices"][lo:hi], g["A_indptr"][c]), shape=(m, n)).toarray()
                qps.append((g["P_diag"][c], g["q"][c], A, g["l"][c], g["u"][c]))
            run("g4_N%d" % N, qps, pool, dict(N=np.array([N])))
        import scenarios
        tr = scenarios.sim_track()
        otrack = M.Track.sim_track()
        for cfgid in (2, 4, 3):
            sc = scenarios.make(cfgid, tr, B=128)
            w = M.Weights.time_optimal() if sc.weights == "time_optimal" else M.Weights.stock()
            lim = M.Limits.stock()
            qps = []
            for i in range(sc.B):
                P, q, A, l, u = M.assemble(otrack, int(sc.wp_id[i]), sc.x0[i], sc.cc_prev[i], sc.lb[i], sc.ub[i], sc.N, w, lim)
                qps.append((np.diag(P), q, A, l, u))
            run("cfg%d" % cfgid, qps, pool, dict(N=np.array([sc.N]), config=np.array([cfgid]), instances=np.array([sc.B])))


if __name__ == "__main__":
    main()
