"""ORACLE, INDEPENDENT LEG (test infrastructure, never shipped, never on the product path).  FROZEN: a change of the
device code must never need a change here.

oracle/osqp_np.py and oracle/osqp_port.c follow the device algorithm step by step (centred start, step indicators,
selective active-set additions, phase 1): their agreement with the kernels shows that three implementations of ONE
heuristic agree.  This module is the leg that shares none of it ("What's weak" 2 and item 3):

  uniqueness_certificate   plain numpy.  A KKT point of a convex QP is AN optimum; the reference's stock weights put no cost
                           on the steering input, on e_psi and on t (src/simulation.py:101-111), so the QP is only positive
                           SEMI-definite and "the" optimum needs a proof.  Every optimal point x' pairs with the multipliers
                           y of any KKT point: rows with y_i != 0 stay on their bound and P x' = P x.  Hence
                               X* is contained in  { x + d :  P d = 0,  A_S d = 0 },   S = equalities and rows with y_i != 0,
                           and the optimum is unique on a coordinate when every vector of null([P; A_S]) vanishes there -
                           no strict complementarity needed (weakly active rows only shrink X* further).
  solve_admm_polish        the restated OSQP iteration (the published ADMM of the solver the reference calls at
                           src/MPC.py:158-159,183; C port, polish = 0, no early attempt, no phase 1) run to eps = 1e-10, then
                           ONE stock OSQP polish: the active set read off that iterate (z - l < -y, u - z < y), one
                           regularised equality-constrained KKT solve, iterative refinement.  No interior point, no
                           indicators, no active-set rounds.  Slow is fine: results are committed as golden G8
                           (tests/golden/make_g8.py) and the tests compare the device with the fixture.
  highs_solution           scipy's bundled HiGHS on the same QP: primal point and objective (its own 1e-7 tolerances).

Reference boundary: the QP is what src/MPC.py:61-155 assembles, `u` is read from dec.x[-2N:] (src/MPC.py:183-189).
"""
from __future__ import annotations

import numpy as np
from scipy import sparse


def _dense(P):
    P = np.asarray(P.toarray() if hasattr(P, "toarray") else P, float)
    return np.diag(P) if P.ndim == 1 else P


def kkt_residuals(P, q, A, l, u, x, y):
    """(primal violation, stationarity, complementarity) of (x, y): plain numpy, inf-norms, unscaled."""
    P, A = _dense(P), np.asarray(A, float)
    Ax = A @ x
    prim = float(np.max(np.maximum(np.maximum(l - Ax, Ax - u), 0.0)))
    stat = float(np.max(np.abs(P @ x + q + A.T @ y)))
    fin_u, fin_l = np.isfinite(u) & (u < 1e20), np.isfinite(l) & (l > -1e20)
    with np.errstate(invalid="ignore"):
        cu = np.where(fin_u, np.maximum(y, 0.0) * np.abs(u - Ax), np.where(y > 0, np.inf, 0.0))
        cl = np.where(fin_l, np.maximum(-y, 0.0) * np.abs(Ax - l), np.where(y < 0, np.inf, 0.0))
    return prim, stat, float(np.max(np.maximum(cu, cl)))


def uniqueness_certificate(P, A, l, u, x, y, coords, tol_y=1e-7, rank_tol=1e-10, tol=1e-6):
    """Is the optimum unique on `coords` (indices into x)?  -> dict(unique, worst, null_dim, smallest_sv, strong_rows)

    S = rows with l = u plus rows whose multiplier exceeds tol_y max(1, |y|_inf) (a smaller |y_i| is treated as zero:
    conservative, S only gets smaller and the null space larger).  worst = largest Euclidean norm of a coordinate's row
    in an orthonormal basis of null([P; A_S]) = how far that coordinate can move per unit step inside the candidate set."""
    P, A = _dense(P), np.asarray(A, float)
    x, y = np.asarray(x, float), np.asarray(y, float)
    fin = np.isfinite(l) & np.isfinite(u) & (l > -1e20) & (u < 1e20)
    with np.errstate(invalid="ignore"):
        eq = fin & ((u - l) <= 1e-12 * np.maximum(1.0, np.abs(np.where(fin, l, 0.0))))
    strong = eq | (np.abs(y) > tol_y * max(1.0, float(np.max(np.abs(y)))))
    M = np.vstack([P, A[strong]])
    _, s, vt = np.linalg.svd(M, full_matrices=True)
    rank = int(np.sum(s > rank_tol * s[0]))
    null = vt[rank:].T                                   # n x (n - rank)
    coords = np.asarray(coords, int)
    worst = float(np.max(np.linalg.norm(null[coords, :], axis=1))) if null.shape[1] else 0.0
    return dict(unique=bool(worst <= tol), worst=worst, null_dim=int(null.shape[1]), smallest_sv=float(s[rank - 1] / s[0]),
                strong_rows=int(strong.sum()))


def compared_coordinates(N):
    """Indices of z = (x_0..x_N, u_0..u_{N-1}) the parity statements compare: everything but the cost-free e_psi_N and
    kappa_{N-1}; and the two entries of the first control (v_0, kappa_0) that get_control returns."""
    keep = np.ones(5 * N + 3, bool)
    keep[[3 * N + 1, 5 * N + 2]] = False
    return np.flatnonzero(keep), np.array([3 * (N + 1), 3 * (N + 1) + 1])


def stock_polish(P, q, A, l, u, x, y, delta=1e-9, refine=10):
    """ONE OSQP polish (Stellato et al. 2020, section 4): active rows read off (x, y) - lower where z - l < -y, upper
    where u - z < y - then  [P + delta I, A_act'; A_act, -delta I] [x; lam] = [-q; b_act]  with `refine` steps of iterative
    refinement against the unregularised system.  -> (x, y, n_active)"""
    P, A = _dense(P), np.asarray(A, float)
    n, m = q.size, l.size
    z = A @ x
    low = (z - l) < -y
    upp = (u - z) < y
    act = np.flatnonzero(low | upp)
    Aa = A[act]
    b = np.where(low[act], l[act], u[act])
    k = act.size
    K = np.block([[P, Aa.T], [Aa, np.zeros((k, k))]])
    Kreg = K + np.diag(np.concatenate([np.full(n, delta), np.full(k, -delta)]))
    rhs = np.concatenate([-q, b])
    import scipy.linalg as sla
    lu = sla.lu_factor(Kreg)
    sol = sla.lu_solve(lu, rhs)
    for _ in range(refine):
        sol = sol + sla.lu_solve(lu, rhs - K @ sol)
    yp = np.zeros(m)
    yp[act] = sol[n:]
    return sol[:n], yp, int(k)


def solve_admm_polish(P, q, A, l, u, eps=1e-10, max_iter=400000, cert_tol=1e-8):
    """restated OSQP (C port, ADMM only) to eps, then ONE stock polish; the polished point is taken if it passes the KKT
    test at cert_tol, else the ADMM point is returned.  -> dict(x, y, status, admm_iters, polished, kkt)"""
    import oracle_c as OC
    st = OC.settings(polish=0, early_polish=0, phase1=0, eps_abs=eps, eps_rel=eps, max_iter=max_iter)
    x, y, info = OC.solve(_dense(P), q, A, l, u, st)
    out = dict(status=int(info.status), admm_iters=int(info.iters), polished=0)
    if info.status not in (1, 2, -2):
        out.update(x=np.full(q.size, np.nan), y=np.full(l.size, np.nan), kkt=(np.nan,) * 3)
        return out
    xp, yp, _ = stock_polish(P, q, A, l, u, x, y)
    kp = kkt_residuals(P, q, A, l, u, xp, yp)
    if max(kp) <= cert_tol:
        out.update(x=xp, y=yp, kkt=kp, polished=1)
    else:
        out.update(x=x, y=y, kkt=kkt_residuals(P, q, A, l, u, x, y))
    return out


def _split_rows(A, l, u):
    """The reference's QP has A = [dynamics rows ; I] (src/MPC.py:128-147): rows with l = u are equalities E x = e, the
    identity rows are bounds lo <= x <= hi.  -> (E, e, eq_rows, lo, hi, bound_row_of_variable)"""
    A = np.asarray(A, float)
    m, n = A.shape
    single = (np.count_nonzero(A, axis=1) == 1) & (np.abs(A).max(axis=1) == 1.0) & (A.max(axis=1) == 1.0)
    brow = np.full(n, -1)
    for i in np.flatnonzero(single):
        j = int(np.argmax(A[i]))
        if brow[j] < 0:
            brow[j] = i
    eq_rows = np.array([i for i in range(m) if i not in set(brow[brow >= 0])], int)
    if np.any(brow < 0) or not np.allclose(l[eq_rows], u[eq_rows]):
        raise ValueError("not the reference's row structure [equalities ; identity]")
    lo = np.where(l[brow] > -1e20, l[brow], -np.inf)
    hi = np.where(u[brow] < 1e20, u[brow], np.inf)
    return A[eq_rows], l[eq_rows].astype(float), eq_rows, lo, hi, brow


def solve_primal_active_set(P, q, A, l, u, x_near=None, tol=1e-9, max_iter=20000):
    """Textbook PRIMAL ACTIVE-SET method for the convex QP (Nocedal & Wright, Numerical Optimization, Alg. 16.3) in dense numpy,
    with the null-space solve of every equality-constrained subproblem by SVD and the inertia-controlling rule for a
    positive SEMI-definite reduced Hessian (a direction of zero curvature with a non-zero gradient is followed to the next
    bound).  No interior point, no step indicators, no scaling, no regularisation: nothing of the device's algorithm.
    Start: the feasible point closest (1-norm) to x_near, by HiGHS' LP solver (scipy.optimize.linprog).
    -> dict(x, y, status (1 solved, -3 infeasible, -2 iteration limit), iters, kkt)"""
    from scipy.optimize import linprog
    P = _dense(P)
    q = np.asarray(q, float)
    l, u = np.asarray(l, float), np.asarray(u, float)
    E, e, eq_rows, lo, hi, brow = _split_rows(A, l, u)
    n, m = q.size, l.size
    x0 = np.zeros(n) if x_near is None else np.asarray(x_near, float)
    # phase 1 (LP): min sum w  s.t.  -w <= x - x0 <= w,  E x = e,  lo <= x <= hi
    I_n = np.eye(n)
    res = linprog(np.concatenate([np.zeros(n), np.ones(n)]), A_ub=np.block([[I_n, -I_n], [-I_n, -I_n]]), b_ub=np.concatenate([x0, -x0]),
                  A_eq=np.hstack([E, np.zeros((E.shape[0], n))]), b_eq=e,
                  bounds=[(a if np.isfinite(a) else None, b if np.isfinite(b) else None) for a, b in zip(lo, hi)] + [(0, None)] * n,
                  method="highs")
    if res.status == 2:
        return dict(x=np.full(n, np.nan), y=np.full(m, np.nan), status=-3, iters=0, kkt=(np.nan,) * 3)
    if res.status != 0:
        raise RuntimeError("phase-1 LP: " + res.message)
    x = np.clip(res.x[:n], lo, hi)
    fixed = np.isfinite(lo) & np.isfinite(hi) & (hi - lo <= 1e-12 * np.maximum(1.0, np.abs(lo)))
    scale = 1.0 + np.max(np.abs(x))
    atl = np.isfinite(lo) & (x - lo <= 1e-9 * scale)
    atu = np.isfinite(hi) & (hi - x <= 1e-9 * scale) & ~atl
    W = atl | atu | fixed                     # working set: variables held on a bound
    side = np.where(atu, 1, -1)               # which bound
    x = np.where(atl | fixed, lo, np.where(atu, hi, x))
    out_status, it = -2, 0
    nu, ybox = np.zeros(E.shape[0]), np.zeros(n)
    for it in range(1, max_iter + 1):
        g = P @ x + q
        F = ~W
        EF = E[:, F]
        # null space of E_F
        if F.any():
            Uf, sf, Vt = np.linalg.svd(EF, full_matrices=True)
            rk = int(np.sum(sf > 1e-12 * max(1.0, sf[0]))) if sf.size else 0
            Z = Vt[rk:].T
        else:
            Z = np.zeros((0, 0))
        p = np.zeros(n)
        ray = False
        if Z.shape[1]:
            H = Z.T @ P[np.ix_(F, F)] @ Z
            r = Z.T @ g[F]
            lam, V = np.linalg.eigh(0.5 * (H + H.T))
            pos = lam > 1e-12 * max(1.0, lam[-1])
            c = V.T @ r
            flat = ~pos & (np.abs(c) > 1e-10 * (1.0 + np.abs(r).max()))
            if flat.any():                    # zero curvature, non-zero slope: a ray of descent
                pz = -(V[:, flat] @ c[flat])
                ray = True
            else:
                pz = -(V[:, pos] @ (c[pos] / lam[pos]))
            p[F] = Z @ pz
        if not ray and np.max(np.abs(p)) <= 1e-11 * scale:
            # stationary on the working set: multipliers  g + E' nu + ybox = 0,  ybox = 0 on the free variables
            if F.any():
                nu = np.linalg.lstsq(EF.T, -g[F], rcond=None)[0]
            else:
                nu = np.zeros(E.shape[0])
            ybox = np.where(W, -(g + E.T @ nu), 0.0)
            bad = W & ~fixed & (((side > 0) & (ybox < -tol)) | ((side < 0) & (ybox > tol)))
            if not bad.any():
                out_status = 1
                break
            j = int(np.argmax(np.where(bad, np.abs(ybox), -1.0)))      # drop the most wrongly signed bound
            W[j] = False
            continue
        # longest step inside the box
        alpha, jblk, sblk = (np.inf if ray else 1.0), -1, 0
        with np.errstate(divide="ignore", invalid="ignore"):
            tu = np.where(F & (p > 1e-14 * scale), (hi - x) / p, np.inf)
            tl = np.where(F & (p < -1e-14 * scale), (lo - x) / p, np.inf)
        ju, jl = int(np.argmin(tu)), int(np.argmin(tl))
        if tu[ju] < alpha:
            alpha, jblk, sblk = max(tu[ju], 0.0), ju, 1
        if tl[jl] < alpha:
            alpha, jblk, sblk = max(tl[jl], 0.0), jl, -1
        if not np.isfinite(alpha):
            raise RuntimeError("the QP is unbounded below along a ray")
        x = x + alpha * p
        if jblk >= 0:
            W[jblk] = True
            side[jblk] = sblk
            x[jblk] = hi[jblk] if sblk > 0 else lo[jblk]
        scale = 1.0 + np.max(np.abs(x))
    y = np.zeros(m)
    y[eq_rows] = nu
    y[brow] = ybox
    return dict(x=x, y=y, status=out_status, iters=it, kkt=kkt_residuals(P, q, A, l, u, x, y))


def highs_solution(P, q, A, l, u, time_limit=10.0):
    """(x, objective) of the QP according to scipy's bundled HiGHS, or (None, nan)."""
    try:
        from scipy.optimize._highspy import _core as hs
        h = hs._Highs()
        h.setOptionValue("output_flag", False)
        h.setOptionValue("time_limit", float(time_limit))
        n, m = q.size, l.size
        lp = hs.HighsLp()
        lp.num_col_, lp.num_row_ = n, m
        lp.col_cost_ = np.asarray(q, float)
        inf = hs.kHighsInf
        lp.col_lower_ = np.full(n, -inf)
        lp.col_upper_ = np.full(n, inf)
        lp.row_lower_ = np.where(np.isfinite(l) & (l > -1e20), l, -inf)
        lp.row_upper_ = np.where(np.isfinite(u) & (u < 1e20), u, inf)
        Ac = sparse.csc_matrix(A)
        lp.a_matrix_.format_ = hs.MatrixFormat.kColwise
        lp.a_matrix_.start_ = Ac.indptr.astype(np.int32)
        lp.a_matrix_.index_ = Ac.indices.astype(np.int32)
        lp.a_matrix_.value_ = Ac.data.astype(float)
        model = hs.HighsModel()
        model.lp_ = lp
        Pc = sparse.triu(sparse.csc_matrix(_dense(P)), format="csc")
        hess = hs.HighsHessian()
        hess.dim_ = n
        hess.format_ = hs.HessianFormat.kTriangular
        hess.start_ = Pc.indptr.astype(np.int32)
        hess.index_ = Pc.indices.astype(np.int32)
        hess.value_ = Pc.data.astype(float)
        model.hessian_ = hess
        if h.passModel(model) != hs.HighsStatus.kOk:
            return None, np.nan
        h.run()
        if h.getModelStatus() != hs.HighsModelStatus.kOptimal:
            return None, np.nan
        return np.array(h.getSolution().col_value, float), float(h.getInfo().objective_function_value)
    except Exception:          # the private binding differs between scipy versions: this leg is optional
        return None, np.nan
