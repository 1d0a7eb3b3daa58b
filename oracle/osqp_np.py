"""ORACLE (test infrastructure, never shipped, never on the product path).

Dense-numpy restatement of the OSQP algorithm, i.e. of the arithmetic the
reference delegates to the un-vendored third-party solver at
  /root/reference/src/MPC.py:158-159,183   (osqp.OSQP().setup(...).solve())
  /root/reference/src/reference_path.py:347-349
OSQP is NOT present in /root/reference nor installable here (README.md:62-68 lists
`osqp` with no version; the .gitignore implies the 0.6.x era).  What is restated is
the published algorithm (Stellato, Banjac, Goulart, Bemporad, Boyd: "OSQP: an
operator splitting solver for quadratic programs", Math. Prog. Comp. 2020) with the
0.6.x default constants: Ruiz equilibration (10 passes) + cost scaling, ADMM with
relaxation alpha=1.6, sigma=1e-6, rho=0.1 (x1e3 on equality rows, 1e-6 on free rows),
residual-balancing rho adaptation, termination / infeasibility certificates every 25
iterations.

PARITY UNPINNED at this boundary: the reference ships no tests, no golden vectors and
no solver; nothing produced by stock OSQP exists to compare against.  What pins the
result instead is solver independent: `kkt_certificate` verifies primal feasibility,
stationarity, dual sign and complementarity of a returned point, which for a convex
QP proves global optimality.

Deliberate deviations from stock OSQP (all documented in DESIGN.md):
  * `adaptive_rho_interval` is a fixed iteration count (stock derives it from
    wall-clock setup/solve time, which is not reproducible).
  * polish.  Stock OSQP's polish is ONE equality-constrained solve on the active set
    guessed from the ADMM iterate.  On this QP family that guess is wrong for most
    instances (measured: accepted for 4/12 config-2 and 0/12 config-4 instances at
    eps=1e-3..1e-5) because the steering channel is nearly flat.  `polish=2` here is:
    warm-started regularised primal-dual interior-point refinement (identifies the
    active set), then the same active-set solve iterated (primal-dual active-set
    updates) until the point passes the KKT certificate.  `polish=1` is stock.

This module is the slow, dense, easy-to-audit twin of oracle/osqp_port.c.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import scipy.linalg as sla

OSQP_INFTY = 1e30
MIN_SCALING = 1e-4
MAX_SCALING = 1e4
RHO_MIN = 1e-6
RHO_MAX = 1e6
RHO_TOL = 1e-4
RHO_EQ_OVER_RHO_INEQ = 1e3

SOLVED = 1
SOLVED_INACCURATE = 2
MAX_ITER_REACHED = -2
PRIMAL_INFEASIBLE = -3
DUAL_INFEASIBLE = -4
UNSOLVED = -10


@dataclasses.dataclass
class Settings:
    rho: float = 0.1
    sigma: float = 1e-6
    alpha: float = 1.6
    eps_abs: float = 1e-3
    eps_rel: float = 1e-3
    eps_prim_inf: float = 1e-4
    eps_dual_inf: float = 1e-4
    max_iter: int = 4000
    check_termination: int = 25
    scaling: int = 10
    adaptive_rho: bool = True
    adaptive_rho_interval: int = 50
    adaptive_rho_tolerance: float = 5.0
    polish: int = 0              # 0 off, 1 stock OSQP polish, 2 certified (IPM + active set)
    delta: float = 1e-6          # stock polish regularisation
    polish_refine_iter: int = 3
    # polish=2 parameters
    ipm_tol: float = 1e-8
    ipm_reg: float = 1e-8
    ipm_max_iter: int = 30
    ipm_diverged: float = 1e2    # interior point stops when mu exceeds this multiple of its smallest value so far
    as_delta: float = 1e-9      # (the device uses 1e-10: fewer refinement solves; a general LDL of this KKT matrix loses
                                #  accuracy there - the plan moves by 1e-5 - so the checker keeps OSQP-polish-like 1e-9)
    as_refine: int = 5
    as_rounds: int = 4
    cert_tol: float = 1e-8
    early_polish: int = 1       # polish=2 only: try the polish after this many ADMM iterations (0 = off)
    early_scaling: int = 1      # Ruiz passes before that attempt; the rest precede the full ADMM run
    phase1: int = 1             # polish=2: what the early attempt cannot certify is first tested for infeasibility
                                # (least-squares phase 1 -> Farkas ray -> PRIMAL_INFEASIBLE) before any full ADMM run
    phase1_theta: float = 1.0   # start value of its slacks / multipliers
    ipm_start_slack: float = 0.1    # the attempt after early_polish ADMM iterations (and the retry from phase 1's
    ipm_start_dual: float = 0.2     # ... mu0 = max(ipm_start_mu, ipm_start_dual * ipm_start_slack * |P x + q|_inf)
    ipm_start_mu: float = 0.01      # point) starts the interior point CENTRED: slacks max(distance to the bound,
                                    # ipm_start_slack), multipliers ipm_start_mu / slack (row space of the scaled
                                    # problem); ipm_start_mu = 0: warm start from the ADMM multipliers as after a full run
    as_add_fraction: float = 0.25   # active-set rounds add only the rows violated by at least this fraction of the worst
                                    # violation (row violation / max |A_r.|: the violation of the scaled variable)
    phase1_eps: float = 1e-6    # OSQP's primal-infeasibility test on phase 1's ray uses this eps: the interior-point ray is
                                # accurate to ~1e-7 (|A'y| / |y|), so the test can be much sharper than eps_prim_inf = 1e-4,
                                # which is calibrated for ADMM's slowly converging dual steps


@dataclasses.dataclass
class Result:
    x: np.ndarray
    y: np.ndarray
    status: int
    iters: int
    pri_res: float
    dua_res: float
    obj: float
    polished: int = 0          # 1 polish accepted / certified, -1 rejected, 0 not run
    rho_updates: int = 0
    rho: float = 0.0
    ipm_iters: int = 0
    as_rounds: int = 0
    x_admm: np.ndarray | None = None
    y_admm: np.ndarray | None = None


def _limit(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.where(v > MAX_SCALING, MAX_SCALING, v)


def _ninf(v):
    return float(np.max(np.abs(v))) if v.size else 0.0


class Workspace:
    """Scaled problem data + factorisation cache (dense)."""

    def __init__(self, P, q, A, l, u, st: Settings):
        self.st = st
        P = np.asarray(P, float)
        A = np.asarray(A, float)
        self.n = n = P.shape[0]
        self.m = m = A.shape[0]
        self.P0, self.q0, self.A0 = P.copy(), np.asarray(q, float).copy(), A.copy()
        # the Python wrapper of OSQP clips the bounds to +-OSQP_INFTY before setup
        self.l0 = np.maximum(np.asarray(l, float), -OSQP_INFTY)
        self.u0 = np.minimum(np.asarray(u, float), OSQP_INFTY)
        self.P, self.q, self.A = P.copy(), self.q0.copy(), A.copy()
        self.l, self.u = self.l0.copy(), self.u0.copy()
        self.D = np.ones(n)
        self.E = np.ones(m)
        self.c = 1.0
        if st.scaling:
            self._scale()
        self.Dinv, self.Einv, self.cinv = 1.0 / self.D, 1.0 / self.E, 1.0 / self.c
        self.rho = st.rho
        self._set_rho_vec()
        self._factor()

    # ---- Ruiz equilibration + cost normalisation ------------------------
    def _scale(self):
        n, m = self.n, self.m
        for _ in range(self.st.scaling):
            Dt = np.maximum(np.max(np.abs(self.P), axis=0), np.max(np.abs(self.A), axis=0))
            Et = np.max(np.abs(self.A), axis=1)
            Dt = 1.0 / np.sqrt(_limit(Dt))
            Et = 1.0 / np.sqrt(_limit(Et))
            self.P = Dt[:, None] * self.P * Dt[None, :]
            self.A = Et[:, None] * self.A * Dt[None, :]
            self.q = Dt * self.q
            self.D *= Dt
            self.E *= Et
            c_tmp = float(np.mean(np.max(np.abs(self.P), axis=0)))
            nq = float(_limit(np.array([_ninf(self.q)]))[0])
            c_tmp = max(c_tmp, nq)
            c_tmp = 1.0 / float(_limit(np.array([c_tmp]))[0])
            self.P *= c_tmp
            self.q *= c_tmp
            self.c *= c_tmp
        self.l = self.E * self.l
        self.u = self.E * self.u

    def _set_rho_vec(self):
        lo_inf = self.l < -OSQP_INFTY * MIN_SCALING
        up_inf = self.u > OSQP_INFTY * MIN_SCALING
        self.ctype = np.where(lo_inf & up_inf, -1,
                              np.where(self.u - self.l < RHO_TOL, 1, 0))
        self.rho_vec = np.where(self.ctype == -1, RHO_MIN,
                                np.where(self.ctype == 1,
                                         RHO_EQ_OVER_RHO_INEQ * self.rho, self.rho))

    def _factor(self):
        n, m = self.n, self.m
        K = np.zeros((n + m, n + m))
        K[:n, :n] = self.P + self.st.sigma * np.eye(n)
        K[:n, n:] = self.A.T
        K[n:, :n] = self.A
        K[n:, n:] = -np.diag(1.0 / self.rho_vec)
        self._lu = sla.lu_factor(K)

    def kkt_solve(self, rhs):
        return sla.lu_solve(self._lu, rhs)

    def unscale(self, x, y):
        return self.D * x, self.E * y * self.cinv


def solve(P, q, A, l, u, settings: Settings | None = None, trace=None) -> Result:
    """OSQP: osqp_solve().  `trace`, if a list, receives (iter, x_scaled, z, y) copies.

    With an early polish attempt (polish=2, early_polish > 0) only `early_scaling` of the Ruiz passes
    come first; what the attempt cannot certify is solved again from a cold start on the problem with
    all `scaling` passes, exactly as OSQP would."""
    st = settings or Settings()
    lo_, up_ = np.asarray(l, float), np.asarray(u, float)
    if np.any(lo_ > up_):
        # an empty interval row: trivially infeasible.  Stock OSQP refuses such data at setup; the build reports it as
        # primal infeasible with a zero ray and the width of the gap as the violation (mpmpc_core.hpp, Solver::run)
        return Result(np.zeros(lo_.size and np.asarray(q).size), np.zeros(lo_.size), PRIMAL_INFEASIBLE, 0,
                      float(np.max(lo_ - up_)), 0.0, 0.0)
    if st.polish == 2 and 0 < st.early_polish < st.max_iter and 0 < st.early_scaling < st.scaling:
        first = _solve(P, q, A, l, u, dataclasses.replace(st, scaling=st.early_scaling, early_scaling=0), trace,
                       stop_after_early=True)
        if first is not None:
            return first
        if trace is not None:
            del trace[:]
        full = _solve(P, q, A, l, u, dataclasses.replace(st, early_polish=0), trace)
        return full
    return _solve(P, q, A, l, u, st, trace)


def _solve(P, q, A, l, u, st: Settings, trace=None, stop_after_early=False):
    w = Workspace(P, q, A, l, u, st)
    n, m = w.n, w.m
    x, y, z = np.zeros(n), np.zeros(m), np.zeros(m)
    status = UNSOLVED
    rho_updates = 0
    it = 0
    info = _info(w, x, z, y)
    dx, dy = np.zeros(n), np.zeros(m)
    while it < st.max_iter:
        it += 1
        x_prev, z_prev = x, z
        rhs = np.concatenate([st.sigma * x_prev - w.q, z_prev - y / w.rho_vec])
        sol = w.kkt_solve(rhs)
        xt = sol[:n]
        zt = z_prev + (sol[n:] - y) / w.rho_vec
        x = st.alpha * xt + (1 - st.alpha) * x_prev
        zr = st.alpha * zt + (1 - st.alpha) * z_prev
        z = np.clip(zr + y / w.rho_vec, w.l, w.u)
        dy = w.rho_vec * (zr - z)
        y = y + dy
        dx = x - x_prev
        if trace is not None:
            trace.append((it, x.copy(), z.copy(), y.copy()))
        can_check = st.check_termination and it % st.check_termination == 0
        if can_check:
            info = _info(w, x, z, y)
            status = _check(w, info, dx, dy, st, approximate=False)
            if status != UNSOLVED:
                break
        if st.polish == 2 and it == st.early_polish and st.early_polish < st.max_iter:
            # the polish only needs a reasonable starting point: try it now; if it cannot certify,
            # the ADMM iteration simply goes on
            early = Result(None, None, UNSOLVED, it, 0.0, 0.0, 0.0, 0, rho_updates, w.rho)
            centred = st.ipm_start_mu > 0.0
            th0 = st.ipm_start_slack if centred else _warm_start_floor(_info(w, x, z, y)["pri_res"])
            if _certified_polish(w, x, y, st, early, th0, st.ipm_start_mu):
                xa, ya = w.unscale(x, y)
                early.x_admm, early.y_admm = xa, ya
                return early
            if st.phase1:
                # not certified: before any long ADMM run, ask whether the problem is infeasible at all
                px, py, pit, cert, feasible = _phase1(w, st)
                early.ipm_iters += pit
                if feasible and _certified_polish(w, px, np.zeros(m), st, early, st.ipm_start_slack if centred else 3e-3,
                                                      st.ipm_start_mu):
                    # phase 1 found the problem FEASIBLE: a second polish attempt from its point (inside every box,
                    # well centred) - the warm-started interior point of the first attempt occasionally jams next to a
                    # degenerate vertex
                    early.x_admm, early.y_admm = w.unscale(x, y)
                    return early
                if cert:
                    early.x, early.y = w.unscale(px, py)          # least-violation point, Farkas ray
                    early.status, early.polished = PRIMAL_INFEASIBLE, 0
                    early.pri_res = kkt_certificate(w.P0, w.q0, w.A0, w.l0, w.u0, early.x, early.y)["prim"]
                    early.x_admm, early.y_admm = w.unscale(x, y)
                    return early
            if stop_after_early:
                return None
        if st.adaptive_rho and st.adaptive_rho_interval and it % st.adaptive_rho_interval == 0:
            if not can_check:
                info = _info(w, x, z, y)
            rho_new = _rho_estimate(w, info, z)
            if rho_new > w.rho * st.adaptive_rho_tolerance or rho_new < w.rho / st.adaptive_rho_tolerance:
                w.rho = rho_new
                w._set_rho_vec()
                w._factor()
                rho_updates += 1
    if status == UNSOLVED:
        info = _info(w, x, z, y)
        status = _check(w, info, dx, dy, st, approximate=False)
        if status == UNSOLVED:
            status = _check(w, info, dx, dy, st, approximate=True)
            if status == UNSOLVED:
                status = MAX_ITER_REACHED
    xs, ys = w.unscale(x, y)
    res = Result(xs, ys, status, it, info["pri_res"], info["dua_res"], _obj(w, xs), 0,
                 rho_updates, w.rho, x_admm=xs, y_admm=ys)
    if status not in (SOLVED, SOLVED_INACCURATE, MAX_ITER_REACHED) or not st.polish:
        return res
    if st.polish == 1:
        pol = _polish_stock(w, x, z, y, st)
        if pol is not None:
            px, py, ppri, pdua = pol
            ok = (ppri < info["pri_res"] and pdua < info["dua_res"]) or \
                 (ppri < info["pri_res"] and info["dua_res"] < 1e-10) or \
                 (pdua < info["dua_res"] and info["pri_res"] < 1e-10)
            if ok:
                xs, ys = w.unscale(px, py)
                res.x, res.y, res.pri_res, res.dua_res, res.obj, res.polished = \
                    xs, ys, ppri, pdua, _obj(w, xs), 1
            else:
                res.polished = -1
        return res
    # polish == 2: interior-point refinement + active-set iterations + certificate
    if not _certified_polish(w, x, y, st, res, _warm_start_floor(info["pri_res"])):
        # not certified (typically a marginally infeasible problem that ADMM at a loose eps calls
        # solved): hand back the ADMM iterate, as stock OSQP would, flagged inaccurate
        res.polished, res.status = -1, SOLVED_INACCURATE
    return res


def _warm_start_floor(pri_res):
    """Floor of the warm-started slacks / multipliers of the interior-point stage: the closer the ADMM
    point is to feasibility, the more its small slacks can be trusted (pri_res / 80 in [3e-4, 3e-3])."""
    return min(3e-3, max(3e-4, 0.0125 * pri_res))


def _certified_polish(w, x, y, st, res, theta=3e-3, mu0=0.0) -> bool:
    """Interior-point refinement from the scaled point (x, y), iterated active-set solve, KKT
    certificate.  On success writes the certified point into `res` and returns True."""
    ipm_tol = st.ipm_tol
    xi, yi = x, y
    for attempt in range(2):
        # (the retry at the tighter tolerance CONTINUES the iteration: slacks and multipliers are kept)
        xi, yi, nit, conv, act = _ipm_refine(w, xi, yi, st, ipm_tol, theta, mu0=mu0 if attempt == 0 else 0.0,
                                             state=act["state"] if attempt else None)
        res.ipm_iters += nit
        if not conv:
            break
        # (the retry is more careful: only the upper half of the violations enters per round)
        out = _active_set_polish(w, act, st, add_fraction=st.as_add_fraction if attempt == 0 else max(st.as_add_fraction, 0.5))
        res.as_rounds += out[3]
        if out[2]:
            xs, ys = w.unscale(out[0], out[1])
            cert = kkt_certificate(w.P0, w.q0, w.A0, w.l0, w.u0, xs, ys)
            if cert["ok_tol"](st.cert_tol):
                res.x, res.y, res.polished, res.status = xs, ys, 1, SOLVED
                res.pri_res, res.dua_res, res.obj = cert["prim"], cert["stat"], cert["obj"]
                return True
        ipm_tol *= 1e-4
    return False


def _obj(w, xs):
    return float(0.5 * xs @ w.P0 @ xs + w.q0 @ xs)


def _info(w: Workspace, x, z, y):
    Ax = w.A @ x
    Px = w.P @ x
    Aty = w.A.T @ y
    rp = Ax - z
    rd = Px + w.q + Aty
    return dict(Ax=Ax, Px=Px, Aty=Aty, rp=rp, rd=rd, z=z,
                pri_res=_ninf(w.Einv * rp), dua_res=w.cinv * _ninf(w.Dinv * rd))


def _check(w: Workspace, info, dx, dy, st: Settings, approximate):
    k = 10.0 if approximate else 1.0
    eps_abs, eps_rel = st.eps_abs * k, st.eps_rel * k
    epi, edi = st.eps_prim_inf * k, st.eps_dual_inf * k
    eps_prim = eps_abs + eps_rel * max(_ninf(w.Einv * info["z"]), _ninf(w.Einv * info["Ax"]))
    eps_dual = eps_abs + eps_rel * w.cinv * max(_ninf(w.Dinv * w.q), _ninf(w.Dinv * info["Aty"]),
                                                 _ninf(w.Dinv * info["Px"]))
    prim_ok = info["pri_res"] < eps_prim
    dual_ok = info["dua_res"] < eps_dual
    prim_inf = (not prim_ok) and _primal_infeasible(w, dy, epi)
    dual_inf = (not dual_ok) and _dual_infeasible(w, dx, edi)
    if prim_ok and dual_ok:
        return SOLVED_INACCURATE if approximate else SOLVED
    if prim_inf:
        return PRIMAL_INFEASIBLE
    if dual_inf:
        return DUAL_INFEASIBLE
    return UNSOLVED


def _primal_infeasible(w: Workspace, dy, eps):
    lo_inf = w.l < -OSQP_INFTY * MIN_SCALING
    up_inf = w.u > OSQP_INFTY * MIN_SCALING
    dy = np.where(up_inf & lo_inf, 0.0,
                  np.where(up_inf, np.minimum(dy, 0.0), np.where(lo_inf, np.maximum(dy, 0.0), dy)))
    nrm = _ninf(w.E * dy)
    if nrm > eps:
        lhs = float(np.sum(w.u * np.maximum(dy, 0.0) + w.l * np.minimum(dy, 0.0)))
        if lhs < -eps * nrm:
            return _ninf(w.Dinv * (w.A.T @ dy)) < eps * nrm
    return False


def _dual_infeasible(w: Workspace, dx, eps):
    nrm = _ninf(w.D * dx)
    if nrm > eps:
        if float(w.q @ dx) < -w.c * eps * nrm:
            if _ninf(w.Dinv * (w.P @ dx)) < w.c * eps * nrm:
                Adx = w.Einv * (w.A @ dx)
                lo_inf = w.l < -OSQP_INFTY * MIN_SCALING
                up_inf = w.u > OSQP_INFTY * MIN_SCALING
                bad = ((~up_inf) & (Adx > eps * nrm)) | ((~lo_inf) & (Adx < -eps * nrm))
                return not bool(np.any(bad))
    return False


def _rho_estimate(w: Workspace, info, z):
    pri = _ninf(info["rp"]) / (max(_ninf(z), _ninf(info["Ax"])) + 1e-10)
    dua = _ninf(info["rd"]) / (max(_ninf(w.q), _ninf(info["Aty"]), _ninf(info["Px"])) + 1e-10)
    est = w.rho * np.sqrt(pri / (dua + 1e-10))
    return float(min(max(est, RHO_MIN), RHO_MAX))


def _polish_stock(w: Workspace, x, z, y, st: Settings):
    """OSQP polish.c: one regularised KKT solve on the guessed active set + refinement."""
    n, m = w.n, w.m
    low = (z - w.l) < -y
    upp = ((w.u - z) < y) & ~low
    rows_l, rows_u = np.flatnonzero(low), np.flatnonzero(upp)
    rows = np.concatenate([rows_l, rows_u])
    Ar = w.A[rows]
    k = rows.size
    K0 = np.zeros((n + k, n + k))
    K0[:n, :n] = w.P
    K0[:n, n:] = Ar.T
    K0[n:, :n] = Ar
    Kr = K0.copy()
    Kr[:n, :n] += st.delta * np.eye(n)
    Kr[n:, n:] -= st.delta * np.eye(k)
    rhs = np.concatenate([-w.q, w.l[rows_l], w.u[rows_u]])
    try:
        lu = sla.lu_factor(Kr)
    except Exception:
        return None
    sol = sla.lu_solve(lu, rhs)
    for _ in range(st.polish_refine_iter):
        sol = sol + sla.lu_solve(lu, rhs - K0 @ sol)
    px = sol[:n]
    py = np.zeros(m)
    py[rows] = sol[n:]
    tmp = w.A @ px + py
    pz = np.clip(tmp, w.l, w.u)
    py = tmp - pz
    rp = w.A @ px - pz
    rd = w.P @ px + w.q + w.A.T @ py
    return px, py, _ninf(w.Einv * rp), w.cinv * _ninf(w.Dinv * rd)


# ---------------------------------------------------------------------------
# polish=2, stage 1: warm-started regularised Mehrotra predictor-corrector on
#     min 1/2 x'Px + q'x   s.t.  l <= Ax <= u     (scaled data)
# rows: free (ignored), equality (l == u), inequality with finite lower / upper side.
# ---------------------------------------------------------------------------
def _row_classes(w: Workspace):
    fin_l = w.l > -OSQP_INFTY * MIN_SCALING
    fin_u = w.u < OSQP_INFTY * MIN_SCALING
    eq = fin_l & fin_u & ((w.u - w.l) <= 1e-12 * np.maximum(1.0, np.abs(w.l)))
    L = fin_l & ~eq
    U = fin_u & ~eq
    return eq, L, U


def _ipm_refine(w: Workspace, x0, y0, st: Settings, tol, theta=3e-3, soft=None, stop=None, mu0=0.0, state=None):
    """`soft` (phase 1, see _phase1): per-row gamma^2 >= 0.  A soft row r reads  l <= (Ax)_r + gamma_r w_r <= u  with
    the cost 1/2 w_r^2 and NO other cost (P, q are taken as zero); w_r = gamma_r (zl_r - zu_r) is eliminated, which
    leaves the same iteration with  (Ax)_r - gamma_r^2 y_r  in place of (Ax)_r in the slack equations and
    gamma_r^2 added to the row's diagonal entry of the reduced KKT matrix.  `stop(x, y)`: optional early exit."""
    n, m = w.n, w.m
    eq, L, U = _row_classes(w)
    beq = w.l
    x = x0.copy()
    Ax = w.A @ x
    P, q = (w.P, w.q) if soft is None else (np.zeros_like(w.P), np.zeros_like(w.q))
    g2 = np.zeros(m) if soft is None else np.where(L | U, soft, 0.0)
    # phase 1: the equality rows are as soft as OSQP's ADMM makes them - RHO_EQ_OVER_RHO_INEQ times the weight of an inequality
    # row's violation (mpmpc_core.hpp: P1_EQ_SOFT)
    eqs = 0.0 if soft is None else 1.0 / RHO_EQ_OVER_RHO_INEQ
    nu = np.where(eq, y0, 0.0)
    sl = np.where(L, np.maximum(Ax - w.l, theta), 1.0)
    su = np.where(U, np.maximum(w.u - Ax, theta), 1.0)
    zl = np.where(L, np.maximum(-y0, theta), 0.0)
    zu = np.where(U, np.maximum(y0, theta), 0.0)
    if mu0 > 0.0:           # centred start: every complementarity product equals mu0, no equality multipliers
        if st.ipm_start_dual > 0.0:     # multipliers commensurate with the dual residual they will have to balance
            mu0 = max(mu0, st.ipm_start_dual * theta * _ninf(w.P @ x + w.q))
        nu = np.zeros(m)
        zl = np.where(L, mu0 / sl, 0.0)
        zu = np.where(U, mu0 / su, 0.0)
    tap_l, tap_u = L & (zl > sl), U & (zu > su)         # (before any step: multiplier above slack)
    if state is not None:
        sl, su, zl, zu, tap_l, tap_u = (a.copy() for a in state)
    nb = max(int(L.sum() + U.sum()), 1)
    reg = st.ipm_reg
    conv = False
    it = 0
    stalled = 0
    mu_min = np.inf
    for it in range(st.ipm_max_iter + 1):
        Ax = w.A @ x
        y = nu + zu - zl
        rd = P @ x + q + w.A.T @ y
        req = np.where(eq, Ax - beq - eqs * nu, 0.0)
        rl = np.where(L, Ax - g2 * (zu - zl) - w.l - sl, 0.0)
        ru = np.where(U, w.u - Ax + g2 * (zu - zl) - su, 0.0)
        mu = (np.sum(sl * zl * L) + np.sum(su * zu * U)) / nb
        res = max(_ninf(rd), _ninf(req), _ninf(rl), _ninf(ru))
        if res < max(tol, 1e-11) and mu < tol:      # (residual floor of double precision: ~1e-13; the retry at
            conv = True                             #  ipm_tol * 1e-4 needs its tolerance on mu only)
            break
        if stop is not None and stop(x, y):
            conv = True
            break
        if it == st.ipm_max_iter:
            break
        # the complementarity measure of a feasible problem falls (nearly) monotonically; on an infeasible one the
        # multipliers blow up within a few iterations (mu jumps by 4-5 orders of magnitude): give up at once, the
        # phase-1 test is the one that can decide such an instance
        if soft is None and (mu > st.ipm_diverged * mu_min or (mu < tol * 1e-3 and res > 1e-5)):
            break           # (... or mu has collapsed far below the tolerance while the residual has not moved)
        mu_min = min(mu_min, mu)
        wt = np.where(L, zl / sl, 0.0) + np.where(U, zu / su, 0.0)
        d = np.where(eq, reg + eqs, np.where(L | U, 1.0 / np.maximum(wt, 1e-300), 1e30))
        dk = d + g2                     # soft rows: gamma^2 on top of the barrier term
        K = np.zeros((n + m, n + m))
        K[:n, :n] = P + reg * np.eye(n)
        K[:n, n:] = w.A.T
        K[n:, :n] = w.A
        K[n:, n:] = -np.diag(dk)
        lu = sla.lu_factor(K)

        def newton(rcl, rcu):
            t = np.where(L, (rcl + zl * rl) / sl, 0.0) - np.where(U, (rcu + zu * ru) / su, 0.0)
            rhs2 = np.where(eq, -req, np.where(L | U, -t * d, 0.0))
            rhs = np.concatenate([-rd, rhs2])
            sol = sla.lu_solve(lu, rhs)
            # one refinement step against the un-regularised Newton matrix
            K0x = P @ sol[:n] + w.A.T @ sol[n:]
            K0y = w.A @ sol[:n] - np.where(eq, eqs, dk) * sol[n:]
            sol = sol + sla.lu_solve(lu, rhs - np.concatenate([K0x, K0y]))
            dx, dyv = sol[:n], sol[n:]
            Adx = w.A @ dx - g2 * dyv
            dsl = np.where(L, Adx + rl, 0.0)
            dsu = np.where(U, -Adx + ru, 0.0)
            dzl = np.where(L, (-rcl - zl * dsl) / sl, 0.0)
            dzu = np.where(U, (-rcu - zu * dsu) / su, 0.0)
            dnu = np.where(eq, dyv, 0.0)
            return dx, dnu, dsl, dsu, dzl, dzu

        def maxstep(v, dv, mask):
            r = np.where(mask & (dv < 0), -v / np.where(dv < 0, dv, -1.0), np.inf)
            return float(r.min()) if r.size else np.inf

        dx, dnu, dsl, dsu, dzl, dzu = newton(sl * zl, su * zu)
        a = min(1.0, maxstep(sl, dsl, L), maxstep(su, dsu, U), maxstep(zl, dzl, L), maxstep(zu, dzu, U))
        mu_aff = (np.sum((sl + a * dsl) * (zl + a * dzl) * L) + np.sum((su + a * dsu) * (zu + a * dzu) * U)) / nb
        sig = (mu_aff / mu) ** 3 if mu > 0 else 0.0
        dx, dnu, dsl, dsu, dzl, dzu = newton(sl * zl - sig * mu + dsl * dzl, su * zu - sig * mu + dsu * dzu)
        a = min(1.0, 0.995 * min(maxstep(sl, dsl, L), maxstep(su, dsu, U), maxstep(zl, dzl, L),
                                 maxstep(zu, dzu, U)))
        stalled = stalled + 1 if a < 1e-6 else 0
        if stalled >= 3:            # steps collapse: infeasible or hopelessly degenerate
            break
        x = x + a * dx
        nu = nu + a * dnu
        # active-set indicators of this step (Tapia): the slack of an active bound shrinks faster than its multiplier
        tap_l, tap_u = L & (dsl * zl < dzl * sl), U & (dsu * zu < dzu * su)
        sl, su, zl, zu = sl + a * dsl, su + a * dsu, zl + a * dzl, zu + a * dzu
    y = nu + zu - zl
    act = dict(eq=eq, low=tap_l, upp=tap_u & ~tap_l, L=L, U=U, state=(sl, su, zl, zu, tap_l, tap_u))
    return x, y, it, conv, act


# ---------------------------------------------------------------------------
# Phase 1: is the QP infeasible?  (polish=2, phase1=1; runs on what the early polish attempt could not certify)
#     min 1/2 |w|^2   s.t.  equality rows as they are,  l <= (Ax)_r + gamma_r w_r <= u  on every other finite row
# (every inequality row soft, nothing else in the cost).  Always feasible when the equality rows are; its optimum is
# zero iff the QP is feasible; and at its optimum the multipliers y satisfy  A'y = 0  and
# u'max(y,0) + l'min(y,0) = -|w|^2 < 0:  a Farkas ray, the certificate OSQP's own infeasibility test asks for
# (is_primal_infeasible), reached by ~10 interior-point iterations on the structured KKT system instead of
# hundreds or thousands of ADMM iterations.  gamma_r = 1: unit weight on the violation of the SCALED ROW - the measure OSQP's
# own ADMM iteration minimises on an infeasible QP (its limit point minimises sum_r rho_r (scaled violation of row r)^2, rho
# the same on all inequality rows), so that the least-violation point is the point the reference's solver call would test
# (round 5; rounds 2 - 4 used gamma_r = max |A_r.| of the scaled row: unit weight on the violation of the scaled variable).
# ---------------------------------------------------------------------------
def farkas_certificate(A, l, u, y, eps=1e-4):
    """Solver-independent check that y proves  {x : l <= Ax <= u}  empty (unscaled data), with OSQP's normalisation:
    |A'y|_inf <= eps |y|_inf  and  u'max(y,0) + l'min(y,0) <= -eps |y|_inf, y's wrong-signed entries on one-sided rows
    projected away first.  Returns dict(ok, support, aty, norm) - support and aty relative to |y|_inf."""
    A = np.asarray(A, float)
    l = np.maximum(np.asarray(l, float), -OSQP_INFTY)
    u = np.minimum(np.asarray(u, float), OSQP_INFTY)
    lo_inf = l < -OSQP_INFTY * MIN_SCALING
    up_inf = u > OSQP_INFTY * MIN_SCALING
    y = np.where(up_inf & lo_inf, 0.0, np.where(up_inf, np.minimum(y, 0.0), np.where(lo_inf, np.maximum(y, 0.0), y)))
    nrm = _ninf(y)
    if not (nrm > 0.0) or not np.all(np.isfinite(y)):
        return dict(ok=False, support=0.0, aty=0.0, norm=nrm)
    support = float(np.sum(u * np.maximum(y, 0.0) + l * np.minimum(y, 0.0))) / nrm
    aty = _ninf(A.T @ y) / nrm
    return dict(ok=bool(support < -eps and aty < eps), support=support, aty=aty, norm=nrm)


def _farkas_values(w: Workspace, dy):
    """the three numbers of OSQP's primal-infeasibility test: |E dy|_inf, u'max(dy,0) + l'min(dy,0), |inv(D) A'dy|_inf"""
    lo_inf = w.l < -OSQP_INFTY * MIN_SCALING
    up_inf = w.u > OSQP_INFTY * MIN_SCALING
    dy = np.where(up_inf & lo_inf, 0.0,
                  np.where(up_inf, np.minimum(dy, 0.0), np.where(lo_inf, np.maximum(dy, 0.0), dy)))
    return (_ninf(w.E * dy), float(np.sum(w.u * np.maximum(dy, 0.0) + w.l * np.minimum(dy, 0.0))),
            _ninf(w.Dinv * (w.A.T @ dy)))


def _phase1(w: Workspace, st: Settings, x0=None):
    """Phase 1 on the scaled problem of `w`.  -> (x, y, iterations, certified infeasible, found feasible).  Two ways to
    an infeasibility verdict:
    (A) OSQP's primal-infeasibility test (unscaled norms) at phase1_eps: any iterate whose ray passes is a
        certificate, the iteration stops at the first one;
    (B) the iteration ran to its converged optimum and that optimum violates a bound by more than cert_tol: the least
        violation is positive, so the problem is infeasible however small the margin - taken when the ray's support is
        negative by at least a hundred times its own residual |A'y|."""
    n, m = w.n, w.m
    soft = np.ones(m)
    stop = lambda x, y: _primal_infeasible(w, y, st.phase1_eps)
    # (two digits beyond the polish's tolerance: for an instance infeasible by a tenth of a millimetre the quantities of
    #  the verdict - the ray's support - are themselves at the 1e-9 level)
    x, y, it, conv, _ = _ipm_refine(w, np.zeros(n) if x0 is None else x0, np.zeros(m), st, min(st.ipm_tol * 1e-2, 1e-11), st.phase1_theta,
                                    soft=soft, stop=stop)
    if _primal_infeasible(w, y, st.phase1_eps):
        return x, y, it, True, False
    if conv:            # the ray test failed on the final iterate: the iteration ended at its converged optimum
        nrm, lhs, res = _farkas_values(w, y)
        xs, ys = w.unscale(x, y)
        prim = kkt_certificate(w.P0, w.q0, w.A0, w.l0, w.u0, xs, ys)["prim"]
        if prim > st.cert_tol and nrm > 0.0 and lhs < 0.0 and lhs < -100.0 * res:
            return x, y, it, True, False
        if not prim > st.cert_tol:
            return x, y, it, False, True          # feasible to tolerance
    return x, y, it, False, False


# ---------------------------------------------------------------------------
# polish=2, stage 2: OSQP's active-set solve, iterated (primal-dual active-set
# updates) with residuals accumulated in extended precision.
# ---------------------------------------------------------------------------
def _active_set_polish(w: Workspace, act, st: Settings, tol=1e-9, add_fraction=None):
    add_fraction = st.as_add_fraction if add_fraction is None else add_fraction
    n, m = w.n, w.m
    eq, low, upp, L, U = act["eq"], act["low"].copy(), act["upp"].copy(), act["L"], act["U"]
    x = np.zeros(n)
    y = np.zeros(m)
    gr = np.maximum(np.max(np.abs(w.A), axis=1), 1e-300) if m else np.ones(0)
    for rnd in range(1, st.as_rounds + 1):
        rows = np.flatnonzero(eq | low | upp)
        k = rows.size
        bound = np.where(upp, w.u, w.l)[rows]
        Ar = w.A[rows]
        K0 = np.zeros((n + k, n + k))
        K0[:n, :n] = w.P
        K0[:n, n:] = Ar.T
        K0[n:, :n] = Ar
        Kr = K0.copy()
        Kr[:n, :n] += st.as_delta * np.eye(n)
        Kr[n:, n:] -= st.as_delta * np.eye(k)
        rhs = np.concatenate([-w.q, bound])
        lu = sla.lu_factor(Kr)
        K0l, rhsl = K0.astype(np.longdouble), rhs.astype(np.longdouble)
        sol = np.zeros(n + k, dtype=np.longdouble)
        for _ in range(st.as_refine + 1):
            sol = sol + sla.lu_solve(lu, np.asarray(rhsl - K0l @ sol, float))
        sol = np.asarray(sol, float)
        x = sol[:n]
        y = np.zeros(m)
        y[rows] = sol[n:]
        Ax = w.A @ x
        viol_l = L & ~low & (Ax < w.l - tol)
        viol_u = U & ~upp & (Ax > w.u + tol)
        bad_l = low & (y > tol)
        bad_u = upp & (y < -tol)
        if not (viol_l.any() or viol_u.any() or bad_l.any() or bad_u.any()):
            return x, y, True, rnd
        # only the violations within as_add_fraction of the worst one are added (see mpmpc_settings::as_add_fraction)
        vl = np.where(viol_l, (w.l - Ax) / gr, 0.0)
        vu = np.where(viol_u, (Ax - w.u) / gr, 0.0)
        thr = add_fraction * max(vl.max(initial=0.0), vu.max(initial=0.0))
        viol_l &= ~(vl < thr)
        viol_u &= ~(vu < thr)
        low = (low & ~bad_l) | viol_l
        upp = ((upp & ~bad_u) | viol_u) & ~low
    return x, y, False, st.as_rounds


def kkt_certificate(P, q, A, l, u, x, y):
    """Solver-independent optimality certificate for  min 1/2 x'Px + q'x  s.t. l<=Ax<=u.

    Returns primal violation, stationarity residual, complementarity (multiplier mass on a
    side that is not tight, y_i > 0 only at the upper bound, y_i < 0 only at the lower) and
    the duality gap, all in the unscaled problem.
    """
    P = np.asarray(P, float)
    A = np.asarray(A, float)
    l = np.maximum(np.asarray(l, float), -OSQP_INFTY)
    u = np.minimum(np.asarray(u, float), OSQP_INFTY)
    Ax = A @ x
    prim = float(np.max(np.maximum(np.maximum(l - Ax, Ax - u), 0.0))) if Ax.size else 0.0
    stat = _ninf(P @ x + q + A.T @ y)
    yp, ym = np.maximum(y, 0.0), np.minimum(y, 0.0)
    up_fin = u < OSQP_INFTY * MIN_SCALING
    lo_fin = l > -OSQP_INFTY * MIN_SCALING
    comp = 0.0
    if Ax.size:
        cu = np.where(up_fin, yp * np.abs(u - Ax), np.where(yp > 0, np.inf, 0.0))
        cl = np.where(lo_fin, -ym * np.abs(Ax - l), np.where(ym < 0, np.inf, 0.0))
        comp = float(max(np.max(cu), np.max(cl)))
    primal_obj = float(0.5 * x @ P @ x + q @ x)
    dual_obj = float(-0.5 * x @ P @ x - np.sum(np.where(up_fin, u, 0.0) * yp)
                     - np.sum(np.where(lo_fin, l, 0.0) * ym))
    out = dict(prim=prim, stat=stat, comp=comp, gap=abs(primal_obj - dual_obj), obj=primal_obj)
    out["ok_tol"] = lambda tol: (prim <= tol and stat <= tol and comp <= tol)
    return out
