"""Small dense QP solver for one-off HOST-side setup problems (the speed profile of
`ReferencePath.compute_speed_profile`, n = n_waypoints - 1, solved once per track).

    min 1/2 x'Px + q'x   s.t.  l <= Ax <= u

Mehrotra predictor-corrector interior point followed by an active-set solve that removes the
barrier bias, so the result is a KKT point to ~1e-12.  This is not the per-timestep hot path (that
one runs on the GPU through libmpmpc.so) and it is not used by the tests as a checker.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

INF = 1e20


def solve_qp(P, q, A, l, u, tol=1e-10, max_iter=80, reg=1e-9):
    P = np.asarray(P, float)
    A = np.asarray(A, float)
    q = np.asarray(q, float)
    l = np.asarray(l, float)
    u = np.asarray(u, float)
    n, m = P.shape[0], A.shape[0]
    fl, fu = l > -INF, u < INF
    eq = fl & fu & (u - l <= 1e-12 * np.maximum(1.0, np.abs(l)))
    L, U = fl & ~eq, fu & ~eq
    x = np.zeros(n)
    nu = np.zeros(m)
    Ax = A @ x
    sl = np.where(L, np.maximum(Ax - l, 1.0), 1.0)
    su = np.where(U, np.maximum(u - Ax, 1.0), 1.0)
    zl, zu = np.where(L, 1.0, 0.0), np.where(U, 1.0, 0.0)
    nb = max(int(L.sum() + U.sum()), 1)

    def step_to_boundary(v, dv, mask):
        r = np.where(mask & (dv < 0), -v / np.where(dv < 0, dv, -1.0), np.inf)
        return float(r.min()) if r.size else np.inf

    converged = False
    for _ in range(max_iter):
        Ax = A @ x
        rd = P @ x + q + A.T @ (nu + zu - zl)
        req = np.where(eq, Ax - l, 0.0)
        rl = np.where(L, Ax - l - sl, 0.0)
        ru = np.where(U, u - Ax - su, 0.0)
        mu = (np.sum(sl * zl * L) + np.sum(su * zu * U)) / nb
        if max(np.abs(rd).max(), np.abs(req).max(initial=0), np.abs(rl).max(initial=0),
               np.abs(ru).max(initial=0)) < tol and mu < tol:
            converged = True
            break
        w = np.where(L, zl / sl, 0.0) + np.where(U, zu / su, 0.0)
        d = np.where(eq, reg, np.where(L | U, 1.0 / np.maximum(w, 1e-300), 1e30))
        K = np.block([[P + reg * np.eye(n), A.T], [A, -np.diag(d)]])
        lu = sla.lu_factor(K)

        def newton(rcl, rcu):
            t = np.where(L, (rcl + zl * rl) / sl, 0.0) - np.where(U, (rcu + zu * ru) / su, 0.0)
            rhs = np.concatenate([-rd, np.where(eq, -req, np.where(L | U, -t * d, 0.0))])
            sol = sla.lu_solve(lu, rhs)
            dx, Adx = sol[:n], A @ sol[:n]
            dsl, dsu = np.where(L, Adx + rl, 0.0), np.where(U, -Adx + ru, 0.0)
            dzl = np.where(L, (-rcl - zl * dsl) / sl, 0.0)
            dzu = np.where(U, (-rcu - zu * dsu) / su, 0.0)
            return dx, np.where(eq, sol[n:], 0.0), dsl, dsu, dzl, dzu

        dx, dnu, dsl, dsu, dzl, dzu = newton(sl * zl, su * zu)
        a = min(1.0, step_to_boundary(sl, dsl, L), step_to_boundary(su, dsu, U),
                step_to_boundary(zl, dzl, L), step_to_boundary(zu, dzu, U))
        mu_aff = (np.sum((sl + a * dsl) * (zl + a * dzl) * L) + np.sum((su + a * dsu) * (zu + a * dzu) * U)) / nb
        sig = (mu_aff / mu) ** 3 if mu > 0 else 0.0
        dx, dnu, dsl, dsu, dzl, dzu = newton(sl * zl - sig * mu + dsl * dzl, su * zu - sig * mu + dsu * dzu)
        a = min(1.0, 0.995 * min(step_to_boundary(sl, dsl, L), step_to_boundary(su, dsu, U),
                                 step_to_boundary(zl, dzl, L), step_to_boundary(zu, dzu, U)))
        x, nu = x + a * dx, nu + a * dnu
        sl, su, zl, zu = sl + a * dsl, su + a * dsu, zl + a * dzl, zu + a * dzu
    y = nu + zu - zl
    # active-set finish: equality-constrained solve on the identified active rows
    low, upp = L & (zl > sl), U & (zu > su)
    for _ in range(10):
        rows = np.flatnonzero(eq | low | upp)
        k = rows.size
        Ar = A[rows]
        K0 = np.block([[P, Ar.T], [Ar, np.zeros((k, k))]])
        Kr = K0 + np.diag(np.concatenate([np.full(n, reg), np.full(k, -reg)]))
        rhs = np.concatenate([-q, np.where(upp, u, l)[rows]])
        lu = sla.lu_factor(Kr)
        sol = np.zeros(n + k)
        for _ in range(6):
            sol = sol + sla.lu_solve(lu, rhs - K0 @ sol)
        xa, ya = sol[:n], np.zeros(m)
        ya[rows] = sol[n:]
        Axa = A @ xa
        vl, vu = L & ~low & (Axa < l - 1e-9), U & ~upp & (Axa > u + 1e-9)
        bl, bu = low & (ya > 1e-9), upp & (ya < -1e-9)
        if not (vl.any() or vu.any() or bl.any() or bu.any()):
            x, y = xa, ya
            break
        low = (low & ~bl) | vl
        upp = ((upp & ~bu) | vu) & ~low
    return x, y, converged
