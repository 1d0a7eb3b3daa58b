"""G5 - solve goldens: for every QP the REFERENCE handed to `osqp.OSQP().setup()` in the G4
captures (tests/golden/g4_assembly_N*.npz, made by make_golden.py from src/MPC.py:61-159), the KKT-certified
optimum z*, multipliers y*, status and certificate residuals as produced by the oracle (oracle/osqp_np.py, polish=2),
plus - where it terminates within its time limit - the objective HiGHS reports for the same QP (scipy's bundled
HiGHS; a sanity cross-check of the optimum, not an oracle: SURVEY 8c).

    python tests/golden/make_g5.py          (system python; needs no reference and no GPU)

Writes tests/golden/g5_solutions_N{3,10,30,50}.npz."""
import os
import sys

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import osqp_np as O  # noqa: E402


def highs_objective(P, q, A, l, u, time_limit=5.0):
    """Objective of the QP according to HiGHS, or nan (not available / no optimal status in time)."""
    try:
        from scipy.optimize._highspy import _core as hs
        h = hs._Highs()
        h.setOptionValue("output_flag", False)
        h.setOptionValue("time_limit", float(time_limit))
        n, m = q.size, l.size
        lp = hs.HighsLp()
        lp.num_col_, lp.num_row_ = n, m
        lp.col_cost_ = q.astype(float)
        inf = hs.kHighsInf
        lp.col_lower_ = np.full(n, -inf)
        lp.col_upper_ = np.full(n, inf)
        lp.row_lower_ = np.where(np.isfinite(l), l, -inf)
        lp.row_upper_ = np.where(np.isfinite(u), u, inf)
        Ac = sparse.csc_matrix(A)
        lp.a_matrix_.format_ = hs.MatrixFormat.kColwise
        lp.a_matrix_.start_ = Ac.indptr.astype(np.int32)
        lp.a_matrix_.index_ = Ac.indices.astype(np.int32)
        lp.a_matrix_.value_ = Ac.data.astype(float)
        model = hs.HighsModel()
        model.lp_ = lp
        Pc = sparse.triu(sparse.csc_matrix(P), format="csc")
        hess = hs.HighsHessian()
        hess.dim_ = n
        hess.format_ = hs.HessianFormat.kTriangular
        hess.start_ = Pc.indptr.astype(np.int32)
        hess.index_ = Pc.indices.astype(np.int32)
        hess.value_ = Pc.data.astype(float)
        model.hessian_ = hess
        if h.passModel(model) != hs.HighsStatus.kOk:
            return np.nan
        h.run()
        if h.getModelStatus() != hs.HighsModelStatus.kOptimal:
            return np.nan
        return float(h.getInfo().objective_function_value)
    except Exception:          # the private binding differs between scipy versions: the cross-check is optional
        return np.nan


def main():
    for N in (3, 10, 30, 50):
        g = np.load(os.path.join(HERE, "g4_assembly_N%d.npz" % N))
        C = g["s"].size
        n, m = 5 * N + 3, 8 * N + 6
        X, Y = np.full((C, n), np.nan), np.full((C, m), np.nan)
        status, iters, ipm = np.zeros(C, np.int32), np.zeros(C, np.int32), np.zeros(C, np.int32)
        cert = np.full((C, 3), np.nan)
        farkas = np.full((C, 2), np.nan)       # infeasible cases: support / |y|, |A'y| / |y| of the stored ray
        obj, obj_highs = np.full(C, np.nan), np.full(C, np.nan)
        for c in range(C):
            lo, hi = g["A_case_ptr"][c], g["A_case_ptr"][c + 1]
            A = sparse.csc_matrix((g["A_data"][lo:hi], g["A_indices"][lo:hi], g["A_indptr"][c]), shape=(m, n)).toarray()
            P = np.diag(g["P_diag"][c])
            q, l, u = g["q"][c], g["l"][c], g["u"][c]
            r = O.solve(P, q, A, l, u, O.Settings(polish=2))
            status[c], iters[c], ipm[c] = r.status, r.iters, r.ipm_iters
            if r.x is not None and r.status in (O.SOLVED, O.SOLVED_INACCURATE, O.MAX_ITER_REACHED):
                X[c], Y[c] = r.x, r.y
                k = O.kkt_certificate(P, q, A, l, u, r.x, r.y)
                cert[c] = k["prim"], k["stat"], k["comp"]
                obj[c] = k["obj"]
                if N <= 30:
                    obj_highs[c] = highs_objective(P, q, A, l, u)
            elif r.status == O.PRIMAL_INFEASIBLE:
                # infeasible capture: the least-violation point of phase 1 and the Farkas ray that proves it
                X[c], Y[c] = r.x, r.y
                f = O.farkas_certificate(A, l, u, r.y, O.Settings().phase1_eps)
                farkas[c] = f["support"], f["aty"]
                assert f["ok"], (N, c, f)
            print("N=%d case %2d: status %2d, %4d ADMM + %2d interior-point iterations, certificate %.1e, obj %.9g (HiGHS %.9g)" %
                  (N, c, r.status, r.iters, r.ipm_iters, np.nanmax(cert[c]) if np.isfinite(cert[c]).any() else np.nan, obj[c], obj_highs[c]))
        np.savez_compressed(os.path.join(HERE, "g5_solutions_N%d.npz" % N), x=X, y=Y, status=status, admm_iters=iters,
                            ipm_iters=ipm, certificate=cert, obj=obj, obj_highs=obj_highs, farkas=farkas,
                            note=np.array(["certified optimum (or, status -3, least-violation point + Farkas ray) of the G4 capture "
                                           "of the same index; oracle/osqp_np.py polish=2, phase1=1"]))


if __name__ == "__main__":
    main()
