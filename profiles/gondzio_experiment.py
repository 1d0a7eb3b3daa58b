"""VERDICT r5 item 4, costed before built: would Gondzio's multiple-centrality correctors pay on BASELINE config 3 (time-optimal
weights, N = 50)?  One corrector = one more KKT solve per interior-point iteration - on the terminal-time kernel K2t ~ +17 .. 25 %
of an iteration (three solves today: predictor, corrector, the Sherman-Morrison vector) - so the iteration count has to fall by
more than that.  This script runs the SAME Mehrotra iteration as the device (restated in dense numpy after oracle/osqp_np.py:
_ipm_refine - same centred start, same step rules, same tolerance, one Ruiz pass) on config-3 instances with 0 / 1 / 2
correctors and prints the iteration counts.  A numpy experiment on the CPU: nothing here is shipped or measured on the device.

    python profiles/gondzio_experiment.py [instances] [config]
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "multi-purpose-mpc_amd", "oracle", "bench_support", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import mpmpc                    # noqa: E402
import mpmpc_testlib as T       # noqa: E402
import osqp_np as O             # noqa: E402
import scenarios                # noqa: E402


def ipm(w, st, tol, correctors, theta, mu0, beta_min=0.1, beta_max=10.0, delta=0.1, gamma=0.1):
    """Mehrotra predictor-corrector of osqp_np._ipm_refine (hard problem, centred start from x = 0) + `correctors` Gondzio
    correctors per iteration.  -> (iterations, converged, KKT solves)"""
    n, m = w.n, w.m
    eq, L, U = O._row_classes(w)
    x = np.zeros(n)
    Ax = w.A @ x
    sl = np.where(L, np.maximum(Ax - w.l, theta), 1.0)
    su = np.where(U, np.maximum(w.u - Ax, theta), 1.0)
    if st.ipm_start_dual > 0.0:
        mu0 = max(mu0, st.ipm_start_dual * theta * O._ninf(w.P @ x + w.q))
    nu = np.zeros(m)
    zl = np.where(L, mu0 / sl, 0.0)
    zu = np.where(U, mu0 / su, 0.0)
    nb = max(int(L.sum() + U.sum()), 1)
    reg = st.ipm_reg
    solves = 0
    for it in range(st.ipm_max_iter + 1):
        Ax = w.A @ x
        y = nu + zu - zl
        rd = w.P @ x + w.q + w.A.T @ y
        req = np.where(eq, Ax - w.l, 0.0)
        rl = np.where(L, Ax - w.l - sl, 0.0)
        ru = np.where(U, w.u - Ax - su, 0.0)
        mu = (np.sum(sl * zl * L) + np.sum(su * zu * U)) / nb
        res = max(O._ninf(rd), O._ninf(req), O._ninf(rl), O._ninf(ru))
        if res < max(tol, 1e-11) and mu < tol:
            return it, True, solves
        if it == st.ipm_max_iter:
            break
        wt = np.where(L, zl / sl, 0.0) + np.where(U, zu / su, 0.0)
        d = np.where(eq, reg, np.where(L | U, 1.0 / np.maximum(wt, 1e-300), 1e30))
        K = np.zeros((n + m, n + m))
        K[:n, :n] = w.P + reg * np.eye(n)
        K[:n, n:] = w.A.T
        K[n:, :n] = w.A
        K[n:, n:] = -np.diag(d)
        lu = sla.lu_factor(K)

        def newton(rcl, rcu, with_residuals=1.0):
            nonlocal solves
            solves += 1
            t = np.where(L, (rcl + with_residuals * zl * rl) / sl, 0.0) - np.where(U, (rcu + with_residuals * zu * ru) / su, 0.0)
            rhs = np.concatenate([-with_residuals * rd, np.where(eq, -with_residuals * req, np.where(L | U, -t * d, 0.0))])
            sol = sla.lu_solve(lu, rhs)
            # one refinement step against the un-regularised Newton matrix (as the oracle's iteration)
            K0x = w.P @ sol[:n] + w.A.T @ sol[n:]
            K0y = w.A @ sol[:n] - np.where(eq, 0.0, d) * sol[n:]
            sol = sol + sla.lu_solve(lu, rhs - np.concatenate([K0x, K0y]))
            dx, dyv = sol[:n], sol[n:]
            Adx = w.A @ dx
            dsl = np.where(L, Adx + with_residuals * rl, 0.0)
            dsu = np.where(U, -Adx + with_residuals * ru, 0.0)
            dzl = np.where(L, (-rcl - zl * dsl) / sl, 0.0)
            dzu = np.where(U, (-rcu - zu * dsu) / su, 0.0)
            return dx, np.where(eq, dyv, 0.0), dsl, dsu, dzl, dzu

        def maxstep(v, dv, mask):
            r = np.where(mask & (dv < 0), -v / np.where(dv < 0, dv, -1.0), np.inf)
            return float(r.min()) if r.size else np.inf

        def step_of(dd):
            _, _, dsl, dsu, dzl, dzu = dd
            return min(maxstep(sl, dsl, L), maxstep(su, dsu, U), maxstep(zl, dzl, L), maxstep(zu, dzu, U))

        aff = newton(sl * zl, su * zu)
        a = min(1.0, step_of(aff))
        mu_aff = (np.sum((sl + a * aff[2]) * (zl + a * aff[4]) * L) + np.sum((su + a * aff[3]) * (zu + a * aff[5]) * U)) / nb
        sig = (mu_aff / mu) ** 3 if mu > 0 else 0.0
        dd = newton(sl * zl - sig * mu + aff[2] * aff[4], su * zu - sig * mu + aff[3] * aff[5])
        a = min(1.0, 0.995 * step_of(dd))
        for _ in range(correctors):
            if a >= 1.0:
                break
            at = min(1.0, a + delta)                    # the step the corrector aims at
            mut = sig * mu
            vl = (sl + at * dd[2]) * (zl + at * dd[4])
            vu = (su + at * dd[3]) * (zu + at * dd[5])

            def target(v, mask):
                t = np.where(v < beta_min * mut, beta_min * mut - v, np.where(v > beta_max * mut, beta_max * mut - v, 0.0))
                return np.where(mask, np.maximum(t, -beta_max * mut), 0.0)
            # corrector: zero residuals, complementarity right-hand side = the projection onto the box around the target
            cc = newton(-target(vl, L), -target(vu, U), with_residuals=0.0)
            trial = tuple(p + c for p, c in zip(dd, cc))
            a2 = min(1.0, 0.995 * step_of(trial))
            if a2 >= a + gamma * delta:
                dd, a = trial, a2
            else:
                break
        x = x + a * dd[0]
        nu = nu + a * dd[1]
        sl, su, zl, zu = sl + a * dd[2], su + a * dd[3], zl + a * dd[4], zu + a * dd[5]
    return st.ipm_max_iter, False, solves


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    config = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    emu = T.Emul()
    tr = scenarios.sim_track()
    sc = scenarios.make(config, tr, B=B)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, tr, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub), obstacles=sc.obstacles)
    dev = mpmpc.default_settings()
    st = O.Settings(polish=2, scaling=1)        # (the early attempt's single Ruiz pass)
    tol = 1e-7 if config != 3 else dev.ipm_tol
    th = 3 * dev.ipm_start_slack if config == 3 else dev.ipm_start_slack          # (K2t's tuned start: mpmpc_reduced_t.hpp)
    m0 = 10 * dev.ipm_start_mu if config == 3 else dev.ipm_start_mu
    ws = []
    for i in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
        ws.append(O.Workspace(np.diag(Pd), q, A, l, u, st))
    print("config %d, N = %d, %d instances, tolerance %.0e (dense numpy restatement of the device's interior point on the full problem)" % (config, sc.N, B, tol))
    base = [ipm(w, st, tol, 0, th, m0) for w in ws]
    b_it, b_so = np.mean([r[0] for r in base]), np.mean([r[2] for r in base])
    print("  Mehrotra alone: iterations mean %.2f (max %d), converged %d / %d, KKT solves per instance %.1f" % (b_it, max(r[0] for r in base), sum(r[1] for r in base), B, b_so))
    for k, delta, bmin, bmax in ((1, 0.1, 0.1, 10.0), (1, 0.3, 0.1, 10.0), (1, 0.5, 0.1, 10.0), (1, 0.3, 0.3, 3.0), (2, 0.1, 0.1, 10.0), (2, 0.3, 0.1, 10.0)):
        r = [ipm(w, st, tol, k, th, m0, beta_min=bmin, beta_max=bmax, delta=delta) for w in ws]
        it, so = np.mean([a[0] for a in r]), np.mean([a[2] for a in r])
        print("  %d corrector(s), delta %.1f, target box [%.1f, %.1f] mu: iterations mean %.2f (max %d), converged %d / %d  ->  iterations %+.1f %%, KKT solves %+.1f %%"
              % (k, delta, bmin, bmax, it, max(a[0] for a in r), sum(a[1] for a in r), B, 100 * (it / b_it - 1), 100 * (so / b_so - 1)))
    print("  (K2t: an iteration = one factorisation + three KKT solves + the step rules; one corrector = one more solve with its Sherman-Morrison"
          " correction, a second step-length rule and two wave reductions: +20 .. 25 % of an iteration)")
