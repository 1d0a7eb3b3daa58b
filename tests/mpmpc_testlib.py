"""Shared helpers for the test-suite (host side only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multi-purpose-mpc_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "bench_support"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import mpmpc  # noqa: E402
import scenarios  # noqa: E402

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def stock_config(N, weights="stock", max_batch=1, circular=True):
    Q, R, QN = scenarios.WEIGHTS[weights]
    return mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                             scenarios.AY_MAX, scenarios.CAR_LENGTH, circular=circular, max_batch=max_batch)


def qp_to_dense(qp_i, N):
    """Stage-blocked fields [27, LD] of one instance -> (Pdiag, q, A, l, u) in the reference's layout."""
    nx = 3
    n = 5 * N + 3
    m = 8 * N + 6
    A = np.zeros((m, n))
    Pd = np.zeros(n)
    q = np.zeros(n)
    l = np.zeros(m)
    u = np.zeros(m)
    ne = nx * (N + 1)
    A[np.arange(ne), np.arange(ne)] = -1.0
    A[ne + np.arange(n), np.arange(n)] = 1.0
    for k in range(N + 1):
        f = qp_i[:, k]
        l[3 * k:3 * k + 3] = u[3 * k:3 * k + 3] = f[4:7]
        xs = slice(3 * k, 3 * k + 3)
        Pd[xs], q[xs] = f[22:25], f[17:20]
        l[ne + 3 * k:ne + 3 * k + 3], u[ne + 3 * k:ne + 3 * k + 3] = f[7:10], f[12:15]
        if k < N:
            us = slice(ne + 2 * k, ne + 2 * k + 2)
            Pd[us], q[us] = f[25:27], f[20:22]
            l[2 * ne + 2 * k:2 * ne + 2 * k + 2], u[2 * ne + 2 * k:2 * ne + 2 * k + 2] = f[10:12], f[15:17]
            ds, a10, a20, b20 = f[0:4]
            r0, c0, cu = 3 * (k + 1), 3 * k, ne + 2 * k
            A[r0, c0], A[r0, c0 + 1] = 1.0, ds
            A[r0 + 1, c0], A[r0 + 1, c0 + 1] = a10, 1.0
            A[r0 + 2, c0], A[r0 + 2, c0 + 2] = a20, 1.0
            A[r0 + 1, cu + 1] = ds
            A[r0 + 2, cu] = b20
    l = np.where(l <= -1e30, -np.inf, l)
    u = np.where(u >= 1e30, np.inf, u)
    return Pd, q, A, l, u


def qp_to_dense_full(qp_i, N, cfg):
    """(P, q, A, l, u) of one instance with the WHOLE weight matrices in P, as the reference builds it (src/MPC.py:150):
    K1's 27 stage fields carry the diagonals of the blocks, the configuration their off-diagonal entries."""
    Pd, q, A, l, u = qp_to_dense(qp_i, N)
    P = np.diag(Pd)
    ne = 3 * (N + 1)
    for k in range(N + 1):
        od = cfg.QN_offdiag if k == N else cfg.Q_offdiag
        for (i, j), v in zip(((0, 1), (0, 2), (1, 2)), od):
            P[3 * k + i, 3 * k + j] = P[3 * k + j, 3 * k + i] = v
        if k < N:
            P[ne + 2 * k, ne + 2 * k + 1] = P[ne + 2 * k + 1, ne + 2 * k] = cfg.R_offdiag[0]
    return P, q, A, l, u


def kkt_batch(qp, N, z, y):
    """Vectorised KKT residuals of a whole batch from the stage-blocked fields qp [27, B, LD]:
    (prim, stat, comp) [B] each, in the reference's unscaled problem (rows [dynamics; state boxes; input
    boxes], src/MPC.py:128-147).  Independent of the kernels' own certificate: plain numpy on the
    K1 output and the returned (z, y)."""
    B = qp.shape[1]
    f = qp[:, :, :N + 1]
    ne = 3 * (N + 1)
    x = z[:, :ne].reshape(B, N + 1, 3)
    u = z[:, ne:].reshape(B, N, 2)
    nu = y[:, :ne].reshape(B, N + 1, 3)
    yx = y[:, ne:2 * ne].reshape(B, N + 1, 3)
    yu = y[:, 2 * ne:].reshape(B, N, 2)
    ds, a10, a20, b20 = (f[i, :, :N] for i in range(4))
    # dynamics rows: -x_k + A_{k-1} x_{k-1} + B_{k-1} u_{k-1} = beq_k
    r = -x.copy()
    r[:, 1:, 0] += x[:, :-1, 0] + ds * x[:, :-1, 1]
    r[:, 1:, 1] += a10 * x[:, :-1, 0] + x[:, :-1, 1] + ds * u[:, :, 1]
    r[:, 1:, 2] += a20 * x[:, :-1, 0] + x[:, :-1, 2] + b20 * u[:, :, 0]
    beq = np.moveaxis(f[4:7], 0, -1)
    lo_x, hi_x = np.moveaxis(f[7:10], 0, -1), np.moveaxis(f[12:15], 0, -1)
    lo_u, hi_u = np.moveaxis(f[10:12, :, :N], 0, -1), np.moveaxis(f[15:17, :, :N], 0, -1)
    viol = lambda v, lo, hi: np.maximum(np.maximum(lo - v, v - hi), 0.0)
    prim = np.maximum(np.abs(r - beq).max(axis=(1, 2)),
                      np.maximum(viol(x, lo_x, hi_x).max(axis=(1, 2)), viol(u, lo_u, hi_u).max(axis=(1, 2))))
    # stationarity: P z + q + A' y
    sx = np.moveaxis(f[22:25], 0, -1) * x + np.moveaxis(f[17:20], 0, -1) - nu + yx
    sx[:, :-1, 0] += nu[:, 1:, 0] + a10 * nu[:, 1:, 1] + a20 * nu[:, 1:, 2]
    sx[:, :-1, 1] += ds * nu[:, 1:, 0] + nu[:, 1:, 1]
    sx[:, :-1, 2] += nu[:, 1:, 2]
    su = np.moveaxis(f[25:27, :, :N], 0, -1) * u + np.moveaxis(f[20:22, :, :N], 0, -1) + yu
    su[:, :, 0] += b20 * nu[:, 1:, 2]
    su[:, :, 1] += ds * nu[:, 1:, 1]
    stat = np.maximum(np.abs(sx).max(axis=(1, 2)), np.abs(su).max(axis=(1, 2)))

    def comp(v, lo, hi, mult):
        with np.errstate(invalid="ignore"):
            return _comp(v, lo, hi, mult)

    def _comp(v, lo, hi, mult):
        up = np.where(hi < 1e20, np.maximum(mult, 0.0) * np.abs(hi - v), np.where(mult > 0, np.inf, 0.0))
        dn = np.where(lo > -1e20, np.maximum(-mult, 0.0) * np.abs(v - lo), np.where(mult < 0, np.inf, 0.0))
        return np.maximum(up, dn).max(axis=(1, 2))
    return prim, stat, np.maximum(comp(x, lo_x, hi_x, yx), comp(u, lo_u, hi_u, yu))


def objective_batch(qp, N, z):
    """1/2 z'Pz + q'z per instance from the stage fields (diagonal weights)."""
    B = qp.shape[1]
    f = qp[:, :, :N + 1]
    ne = 3 * (N + 1)
    x = z[:, :ne].reshape(B, N + 1, 3)
    u = z[:, ne:].reshape(B, N, 2)
    Px, qx = np.moveaxis(f[22:25], 0, -1), np.moveaxis(f[17:20], 0, -1)
    Pu, qu = np.moveaxis(f[25:27, :, :N], 0, -1), np.moveaxis(f[20:22, :, :N], 0, -1)
    return (0.5 * Px * x * x + qx * x).sum(axis=(1, 2)) + (0.5 * Pu * u * u + qu * u).sum(axis=(1, 2))


def controls_vs_reference(qp, N, sol, ref, tol=1e-6):
    """Controls of the device against the oracle's on the instances both solved.  The reference's stock weights put NO
    cost on the steering input (R = diag(0.5, 0), src/simulation.py:104) and none on e_psi and t: the QP is positive
    SEMI-definite, and where a corridor bound is weakly active (zero multiplier) its optimum is a face, not a point -
    two correct solvers may return different points of it.  Such an instance counts as an ALTERNATIVE OPTIMUM when
    both points carry a KKT certificate (1e-8) and their objectives agree to 1e-9 relative; everything else must agree
    to `tol`.  -> (worst |u - u_ref| over the rest, indices of the alternative optima)"""
    both = (sol.status == 1) & (ref["status"] == 1)
    d = np.abs(sol.u0 - ref["u0"]).max(axis=1)
    far = np.flatnonzero(both & (d > tol))
    alt = []
    if far.size:
        q = qp[:, far, :]
        od, orf = objective_batch(q, N, sol.z[far]), objective_batch(q, N, ref["z"][far])
        cd = np.max(kkt_batch(q, N, sol.z[far], sol.y[far]), axis=0)
        cr = np.max(kkt_batch(q, N, ref["z"][far], ref["y"][far]), axis=0)
        same = (np.abs(od - orf) <= 1e-9 * np.maximum(1.0, np.abs(orf))) & (cd <= 1e-8) & (cr <= 1e-8)
        alt = far[same]
    rest = both.copy()
    rest[alt] = False
    return (float(d[rest].max()) if rest.any() else 0.0), np.asarray(alt, int)


def farkas_batch(qp, N, y, eps=1e-6):
    """Vectorised check that the rays y [B, 8N+6] prove their QPs infeasible (OSQP's primal-infeasibility criterion,
    plain numpy on K1's stage fields qp [27, B, LD]; eps = the library's phase1_eps; rows [dynamics; state boxes; input boxes] as in
    src/MPC.py:128-147):  |A'y|_inf <= eps |y|_inf  and  u'max(y,0) + l'min(y,0) <= -eps |y|_inf, with no multiplier
    mass on an infinite side.  -> (ok [B], support / |y| [B], |A'y| / |y| [B])"""
    B = qp.shape[1]
    f = qp[:, :, :N + 1]
    ne = 3 * (N + 1)
    nu = y[:, :ne].reshape(B, N + 1, 3)
    yx = y[:, ne:2 * ne].reshape(B, N + 1, 3)
    yu = y[:, 2 * ne:].reshape(B, N, 2)
    ds, a10, a20, b20 = (f[i, :, :N] for i in range(4))
    beq = np.moveaxis(f[4:7], 0, -1)
    lo_x, hi_x = np.moveaxis(f[7:10], 0, -1), np.moveaxis(f[12:15], 0, -1)
    lo_u, hi_u = np.moveaxis(f[10:12, :, :N], 0, -1), np.moveaxis(f[15:17, :, :N], 0, -1)
    # A'y per variable (the stationarity expression of kkt_batch without P z + q)
    sx = -nu + yx
    sx[:, :-1, 0] += nu[:, 1:, 0] + a10 * nu[:, 1:, 1] + a20 * nu[:, 1:, 2]
    sx[:, :-1, 1] += ds * nu[:, 1:, 0] + nu[:, 1:, 1]
    sx[:, :-1, 2] += nu[:, 1:, 2]
    su = yu.copy()
    su[:, :, 0] += b20 * nu[:, 1:, 2]
    su[:, :, 1] += ds * nu[:, 1:, 1]
    nrm = np.maximum(np.abs(y).max(axis=1), 1e-300)
    aty = np.maximum(np.abs(sx).max(axis=(1, 2)), np.abs(su).max(axis=(1, 2))) / nrm

    def side(mult, lo, hi):
        fin_hi, fin_lo = hi < 1e20, lo > -1e20
        wrong = ((mult > 0) & ~fin_hi) | ((mult < 0) & ~fin_lo)
        val = np.where(fin_hi, hi, 0.0) * np.maximum(mult, 0.0) + np.where(fin_lo, lo, 0.0) * np.minimum(mult, 0.0)
        return val.sum(axis=(1, 2)), wrong.any(axis=(1, 2))
    s1, w1 = side(yx, lo_x, hi_x)
    s2, w2 = side(yu, lo_u, hi_u)
    support = ((beq * nu).sum(axis=(1, 2)) + s1 + s2) / nrm
    ok = (support <= -eps) & (aty <= eps) & ~w1 & ~w2 & np.isfinite(y).all(axis=1)
    return ok, support, aty


# (the ctypes view of the kernels' CPU emulation and the handle stand-ins built on it live in bench_support/emulation.py:
#  bench.py --dry-run drives them too, and a benchmark script does not import from tests/)
from emulation import Emul, EmuBackend, DryHandle  # noqa: E402,F401


_WIDE_TRACKS = {}


def wide_track(track, emu, n_cols):
    """The golden track with corridor tables of n_cols > 50 columns (horizons above 50: golden G3 holds 50), built by the CPU
    emulation of the device's corridor kernels from golden G1's grids - bit-identical to G3 on the first 50 columns (checked
    here; tests/test_corridor.py pins the emulation to G3 for every start waypoint)."""
    import dataclasses
    key = (id(track), n_cols)
    if key in _WIDE_TRACKS:
        return _WIDE_TRACKS[key]
    g1 = np.load(os.path.join(ROOT, "tests", "golden", "g1_path_sim_track.npz"))
    g3 = np.load(os.path.join(ROOT, "tests", "golden", "g3_corridor.npz"))
    h, w = g1["grid_shape"]
    sm = float(g3["safety_margin"][0])
    n = g1["x"].size
    arrs = [np.ascontiguousarray(g1[k], float) for k in ("x", "y", "psi", "ds_next")]
    bu, bl = np.ascontiguousarray(g1["border_ub"], float), np.ascontiguousarray(g1["border_lb"], float)
    out = {}
    for name in ("free", "obstacles"):
        grid = np.ascontiguousarray(np.unpackbits(g1["grid_" + name])[:h * w].reshape(h, w).astype(np.int8))
        ub, lb = np.zeros((n, n_cols)), np.zeros((n, n_cols))
        bad = emu.lib.emu_corridor(C.c_int(grid.shape[0]), C.c_int(grid.shape[1]), grid.ctypes.data_as(C.POINTER(C.c_int8)),
                                   C.c_double(-1.0), C.c_double(-2.0), C.c_double(0.005), C.c_int(n), *[_d(a) for a in arrs], C.c_int(1),
                                   _d(bu), _d(bl), C.c_int(n_cols), C.c_double(2 * sm), C.c_double(sm), _d(ub), _d(lb), None)
        assert bad == 0 and np.array_equal(ub[:, :50], g3["ub_" + name]) and np.array_equal(lb[:, :50], g3["lb_" + name])
        out["ub_" + name], out["lb_" + name] = ub, lb
    _WIDE_TRACKS[key] = dataclasses.replace(track, **out)
    return _WIDE_TRACKS[key]


def emu_speed_profile(li, kappa, limits, eps=1e-12, device=0):
    """CPU emulation of mpmpc_speed_profile_kernel (same signature as mpmpc.speed_profile)."""
    lib = Emul().lib
    li = np.atleast_2d(np.ascontiguousarray(li, float))
    kappa = np.atleast_2d(np.ascontiguousarray(kappa, float))
    B, n = li.shape
    limits = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(limits, float)), (B, 5)))
    v, status, iters = np.zeros((B, n)), np.zeros(B, np.int32), np.zeros(B, np.int32)
    for p in range(B):
        it = C.c_int(0)
        status[p] = lib.emu_speed_profile(C.c_int(n), _d(li[p]), _d(kappa[p]), _d(limits[p]), C.c_double(eps), _d(v[p]),
                                          C.byref(it))
        iters[p] = it.value
    return v, status, iters


# ---- the independent leg (oracle/independent.py, golden G8): nothing below shares code with the device algorithm
def g8(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "g8_independent_%s.npz" % name))


def compare_with_independent(sol, g, N, tol=1e-6):
    """Device / emulation answers against golden G8 (restated OSQP ADMM to 1e-10 + ONE stock polish; never touched by a
    device commit).  Compared where the independent leg certified ITS OWN point (polished = 1: KKT <= 1e-8): the first
    control to `tol` in (v_0, kappa_0 -> delta_0 through atan), the plan without the cost-free kappa_{N-1} / e_psi_N to
    `tol`.  -> dict(compared, worst_u0, worst_plan, verdict_disagreements)"""
    import independent as I
    keep, u0c = I.compared_coordinates(N)
    n = g["x"].shape[0]
    both = (g["polished"] == 1) & (sol.status[:n] == 1)
    du = np.abs(sol.z[:n][both][:, u0c] - g["x"][both][:, u0c])
    dp = np.abs(sol.z[:n][both][:, keep] - g["x"][both][:, keep])
    # verdicts: the ADMM leg's own infeasibility verdict (-3) against a usable / refused answer of the device
    dev_refused = (sol.status[:n] == -3)
    ind_refused = (g["status"] == -3)
    return dict(compared=int(both.sum()), worst_u0=float(du.max()) if both.any() else 0.0, worst_plan=float(dp.max()) if both.any() else 0.0,
                refused_by_both=int((dev_refused & ind_refused).sum()), refused_by_device_only=int((dev_refused & ~ind_refused).sum()),
                refused_by_independent_only=int((~dev_refused & ind_refused).sum()))


def uniqueness_count(qp, N, z, y, idx):
    """oracle/independent.py:uniqueness_certificate on the device's (z, y) of the instances `idx`, on the coordinates the
    parity statements compare.  -> (number certified unique, number checked, worst coordinate freedom)"""
    import independent as I
    keep, _ = I.compared_coordinates(N)
    good, worst = 0, 0.0
    for i in idx:
        Pd, q, A, l, u = qp_to_dense(qp[:, i, :], N)
        c = I.uniqueness_certificate(Pd, A, l, u, z[i], y[i], keep)
        good += int(c["unique"])
        worst = max(worst, c["worst"])
    return good, len(idx), worst


def relaxed_plan_check(qp_i, N, z, y, viol):
    """A status-2 plan (mpmpc_settings::phase1_accept: an instance infeasible by less than OSQP's own primal tolerance) is
    returned as the optimum over boxes widened by 1.5 x the least violation.  Independent statement, plain numpy on K1's
    stage fields: with every box relaxed to exactly what z itself uses - lo' = min(lo, z), hi' = max(hi, z) - the returned
    (z, y) is a KKT point of that QP (oracle/independent.py:kkt_residuals), it is unique there on the compared coordinates,
    and no box was relaxed by more than 1.5 x resid[0].  -> dict(kkt, unique, worst_relaxation / resid0)"""
    import independent as I
    Pd, q, A, l, u = qp_to_dense(qp_i, N)
    Az = A @ z
    lr, ur = np.minimum(l, Az), np.maximum(u, Az)
    ne = 3 * (N + 1)
    lr[:ne], ur[:ne] = l[:ne], u[:ne]                      # (the dynamics rows are not relaxed)
    with np.errstate(invalid="ignore"):
        relax = float(max(np.max(np.where(np.isfinite(l), l - lr, 0.0)), np.max(np.where(np.isfinite(u), ur - u, 0.0))))
    keep, _ = I.compared_coordinates(N)
    kkt = I.kkt_residuals(Pd, q, A, lr, ur, z, y)
    uq = I.uniqueness_certificate(Pd, A, lr, ur, z, y, keep)
    return dict(kkt=max(kkt), unique=bool(uq["unique"]), relaxation=relax, ratio=relax / viol if viol > 0 else np.inf)


def branch_compare(cfgid, B, settings=None):
    """Which branch of src/MPC.py:185-216 does get_control take - a fresh plan or the fallback - with the device's settings,
    against the restated stock OSQP (the C oracle at OSQP's defaults: the arithmetic of the reference's own solver call,
    src/MPC.py:159,183)?  Needs a device.  -> dict(agreement, rows, ...): rows = (instance, device status, device violation,
    stock status, stock iterations, stock pri_res) of every instance on which the two disagree.  (profiles/branch_agreement.py
    prints it; tests/test_gpu_parity.py pins size, direction and cause.)"""
    import oracle_c as OC
    track = scenarios.sim_track()
    sc = scenarios.make(cfgid, track, B=B)
    cfg = stock_config(sc.N, sc.weights, max_batch=B)
    st = settings or mpmpc.default_settings()
    h = mpmpc.Handle(cfg, st)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    h.close()
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN, scenarios.XMAX, 4.0, 0.12)
    stock = OC.mpc_batch(ocfg, OC.settings(polish=0, early_polish=0, phase1=0), track.kappa, track.v_ref, track.ds_next, sc.wp_id,
                         sc.x0, sc.cc_prev, sc.lb, sc.ub)

    def usable(s):
        return np.isin(s, (1, 2, -2))
    dis = np.flatnonzero(usable(sol.status) != usable(stock["status"]))
    rows = [(int(i), int(sol.status[i]), float(sol.resid[i, 0]), int(stock["status"][i]), int(stock["iters"][i, 0]), float(stock["resid"][i, 0]))
            for i in dis]
    return dict(agreement=1.0 - dis.size / B, rows=rows, device=dict(zip(*map(lambda a: a.tolist(), np.unique(sol.status, return_counts=True)))),
                stock=dict(zip(*map(lambda a: a.tolist(), np.unique(stock["status"], return_counts=True)))), B=B,
                threshold=1e-3 + 1e-3 * float(scenarios.UMAX[1]))
