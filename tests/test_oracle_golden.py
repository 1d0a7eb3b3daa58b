"""The oracle's assembly half against what the REFERENCE itself produced (tests/golden/g4_*.npz,
captured from src/MPC.py:61-159 by tests/golden/make_golden.py)."""
import numpy as np
import pytest
from scipy import sparse

import mpc_np as M
import osqp_np as O


def _ulp_close(a, b, ulps=4):
    fin = np.isfinite(a) & np.isfinite(b)
    ok_inf = np.array_equal(a[~fin], b[~fin])
    return ok_inf and np.all(np.abs(a[fin] - b[fin]) <= ulps * np.spacing(np.maximum(np.abs(a[fin]), np.abs(b[fin]))))


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_assembly_matches_reference_capture(N, otrack):
    g = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    w = M.Weights.time_optimal() if str(g["weights"][0]) == "time_optimal" else M.Weights.stock()
    lim = M.Limits.stock()
    assert g["s"].size >= 16
    for c in range(g["s"].size):
        wp = int(g["wp_id"][c])
        # a2: waypoint localisation, a3: t2s
        assert M.current_waypoint(otrack.segment_lengths, g["s"][c]) == wp
        x0 = np.array(M.t2s(*g["pose"][c], otrack.x[wp], otrack.y[wp], otrack.psi[wp]))
        assert np.array_equal(x0, g["x0"][c])
        P, q, A, l, u = M.assemble(otrack, wp, g["x0"][c], g["cc_prev"][c], g["lb"][c], g["ub"][c], N, w, lim)
        lo, hi = g["A_case_ptr"][c], g["A_case_ptr"][c + 1]
        Aref = sparse.csc_matrix((g["A_data"][lo:hi], g["A_indices"][lo:hi], g["A_indptr"][c]), shape=A.shape)
        assert np.array_equal(A, Aref.toarray())                      # bit-exact, pattern included
        assert sparse.csc_matrix(A).nnz == Aref.nnz
        assert np.array_equal(np.diag(P), g["P_diag"][c]) and np.count_nonzero(P) == np.count_nonzero(np.diag(P))
        assert np.array_equal(q, g["q"][c]) and np.array_equal(np.signbit(q), np.signbit(g["q"][c]))
        assert np.array_equal(l, g["l"][c])
        # u: bit-exact except the speed-cap entries, which go through libm tan() (numpy 1.26 in the
        # capture vs numpy 2.x here differ by 1 ulp on a few arguments)
        assert _ulp_close(u, g["u"][c], 4)
        cap = np.flatnonzero(u != g["u"][c])
        assert all((i >= 6 * (N + 1)) and ((i - 6 * (N + 1)) % 2 == 0) for i in cap)


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_assembly_with_full_weights_matches_reference_capture(N, otrack):
    """G4f: NON-diagonal Q, R, QN through the reference itself (tests/golden/make_golden.py assembly_full).  The reference
    puts the whole matrices into P (src/MPC.py:150) and only diag(Q), diag(R) - but the whole QN - into q (src/MPC.py:153-155):
    the numpy restatement reproduces its (P, q, A, l, u) bit for bit, P's sparsity pattern included."""
    g = np.load(M.GOLDEN + "/g4f_assembly_N%d.npz" % N)
    assert str(g["weights"][0]) == "full" and g["s"].size >= 12
    w = full_weights()
    lim = M.Limits.stock()
    n = 5 * N + 3
    assert int(g["P_nnz"][0]) == 9 * (N + 1) + 4 * N                      # dense 3 x 3 and 2 x 2 blocks
    for c in range(g["s"].size):
        wp = int(g["wp_id"][c])
        P, q, A, l, u = M.assemble(otrack, wp, g["x0"][c], g["cc_prev"][c], g["lb"][c], g["ub"][c], N, w, lim)
        Pref = sparse.coo_matrix((g["P_val"][c], (g["P_row"][c], g["P_col"][c])), shape=(n, n)).toarray()
        assert np.array_equal(P, Pref) and np.count_nonzero(P) == int(g["P_nnz"][c])
        lo, hi = g["A_case_ptr"][c], g["A_case_ptr"][c + 1]
        Aref = sparse.csc_matrix((g["A_data"][lo:hi], g["A_indices"][lo:hi], g["A_indptr"][c]), shape=A.shape)
        assert np.array_equal(A, Aref.toarray())
        assert np.array_equal(q, g["q"][c]) and np.array_equal(np.signbit(q), np.signbit(g["q"][c]))
        assert np.array_equal(l, g["l"][c]) and _ulp_close(u, g["u"][c], 4)
        # the quirk: q is NOT -P xr - the off-diagonal entries of Q never reach it
        ey_ref = np.zeros(n)
        ey_ref[3:3 * (N + 1):3] = (g["lb"][c] + g["ub"][c]) / 2
        if np.any(ey_ref[:3 * N] != 0):
            assert not np.allclose(q[:3 * N], -(Pref @ ey_ref)[:3 * N])


def full_weights():
    """the weight set of golden G4f (tests/golden/make_golden.py: FULL_WEIGHTS)"""
    return M.Weights(np.array([[1.0, 0.2, 0.05], [0.2, 0.3, -0.1], [0.05, -0.1, 0.2]]), np.array([[0.5, 0.1], [0.1, 0.2]]),
                     np.array([[1.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 0.4]]))


def test_edge_cases_present():
    g = np.load(M.GOLDEN + "/g4_assembly_N30.npz")
    assert 0 in g["wp_id"] and 199 in g["wp_id"]          # int kappa at wp 0, the 199 -> 0 wrap
    assert np.any(np.all(g["cc_prev"] == 0, axis=1)) and np.any(np.any(g["cc_prev"] != 0, axis=1))
    nnz = np.diff(g["A_case_ptr"])
    assert nnz.min() < nnz.max()                          # kappa == 0 stages drop entries from the pattern


def test_kappa_pred_quirk():
    cc = np.arange(1.0, 13.0) / 20.0                      # N = 6
    kp = M.kappa_pred(cc, 0.12)
    assert kp.shape == (2 * 6 - 3,)
    assert np.array_equal(kp, np.tan(cc[3:] + cc[-1]) / 0.12)


def test_speed_profile_qp_certified():
    """G2: the reference's speed-profile QP (reference_path.py:289-354) and its certified optimum."""
    g2 = np.load(M.GOLDEN + "/g2_speed_profile.npz")
    A = sparse.coo_matrix((g2["A_val"], (g2["A_row"], g2["A_col"])), shape=tuple(g2["A_shape"])).toarray()
    P = np.diag(g2["P_diag"])
    c = O.kkt_certificate(P, g2["q"], A, g2["l"], g2["u"], g2["x"], g2["y"])
    assert c["ok_tol"](1e-9)
    r = O.solve(P, g2["q"], A, g2["l"], g2["u"], O.Settings(polish=2))
    assert r.status == O.SOLVED and r.polished == 1
    assert np.max(np.abs(r.x - g2["x"])) < 1e-9
    assert np.array_equal(g2["v_ref"][:-1], g2["x"]) and g2["v_ref"][-1] == g2["v_ref"][-2]


def test_osqp_restatement_statuses(otrack):
    """Feasible -> solved + certificate; an impossible corridor -> primal infeasible."""
    N = 10
    lim, w = M.Limits.stock(), M.Weights.stock()
    lb, ub = np.full(N, -0.1), np.full(N, 0.1)
    P, q, A, l, u = M.assemble(otrack, 5, np.array([0.01, 0.05, 0.0]), np.zeros(2 * N), lb, ub, N, w, lim)
    r = O.solve(P, q, A, l, u, O.Settings(polish=2))
    assert r.status == O.SOLVED and r.polished == 1
    assert O.kkt_certificate(P, q, A, l, u, r.x, r.y)["ok_tol"](1e-8)
    lb2, ub2 = lb.copy(), ub.copy()
    lb2[0], ub2[0] = 0.09, 0.1                             # unreachable from e_y = 0.01 in one step
    P, q, A, l, u = M.assemble(otrack, 5, np.array([0.01, 0.0, 0.0]), np.zeros(2 * N), lb2, ub2, N, w, lim)
    r = O.solve(P, q, A, l, u, O.Settings())
    assert r.status == O.PRIMAL_INFEASIBLE


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_g5_solutions_are_kkt_points_of_the_reference_qps(N):
    """G5 (tests/golden/make_g5.py): the stored optimum of every QP the reference handed to osqp.setup (G4) passes the
    solver-independent KKT certificate on THAT data, and agrees with the objective HiGHS reports where HiGHS
    terminated (a sanity cross-check only, SURVEY 8c: HiGHS stops at its 1e-7 feasibility tolerance, so its
    objective sits a few 1e-6 on either side of the optimum)."""
    g4 = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    g5 = np.load(M.GOLDEN + "/g5_solutions_N%d.npz" % N)
    n, m = 5 * N + 3, 8 * N + 6
    assert np.sum(g5["status"] == 1) >= 15 and np.any(g5["status"] == -3)       # feasible and infeasible captures
    for c in np.flatnonzero(g5["status"] == 1):
        lo, hi = g4["A_case_ptr"][c], g4["A_case_ptr"][c + 1]
        A = sparse.csc_matrix((g4["A_data"][lo:hi], g4["A_indices"][lo:hi], g4["A_indptr"][c]), shape=(m, n)).toarray()
        k = O.kkt_certificate(np.diag(g4["P_diag"][c]), g4["q"][c], A, g4["l"][c], g4["u"][c], g5["x"][c], g5["y"][c])
        assert k["ok_tol"](1e-8), (N, c, k["prim"], k["stat"], k["comp"])
        assert abs(k["obj"] - g5["obj"][c]) <= 1e-12 * max(1.0, abs(k["obj"]))
        if np.isfinite(g5["obj_highs"][c]):
            assert abs(k["obj"] - g5["obj_highs"][c]) <= 2e-5 * max(1.0, abs(k["obj"]))


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_g5_infeasible_captures_carry_a_farkas_ray(N):
    """Every capture G5 calls infeasible comes with a ray y that PROVES it on the reference's own (A, l, u):
    |A'y| <= eps |y| and u'max(y,0) + l'min(y,0) <= -eps |y| (OSQP's criterion at the oracle's phase1_eps = 1e-6) - checked here with
    plain numpy on the G4 data, independent of any solver.  And no capture is left in between: certified optimum or
    certified infeasible."""
    g4 = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    g5 = np.load(M.GOLDEN + "/g5_solutions_N%d.npz" % N)
    n, m = 5 * N + 3, 8 * N + 6
    assert set(np.unique(g5["status"])) <= {1, -3}
    for c in np.flatnonzero(g5["status"] == -3):
        lo, hi = g4["A_case_ptr"][c], g4["A_case_ptr"][c + 1]
        A = sparse.csc_matrix((g4["A_data"][lo:hi], g4["A_indices"][lo:hi], g4["A_indptr"][c]), shape=(m, n)).toarray()
        y = g5["y"][c]
        l = np.maximum(g4["l"][c], -1e30)
        u = np.minimum(g4["u"][c], 1e30)
        # multipliers on an infinite side must vanish (a one-sided row only pushes one way)
        assert np.all(y[u >= 1e26] <= 0) and np.all(y[l <= -1e26] >= 0)
        nrm = np.max(np.abs(y))
        assert np.max(np.abs(A.T @ y)) <= 1e-6 * nrm
        assert np.sum(u * np.maximum(y, 0) + l * np.minimum(y, 0)) <= -1e-6 * nrm
        f = O.farkas_certificate(A, g4["l"][c], g4["u"][c], y, 1e-6)
        assert f["ok"] and abs(f["support"] - g5["farkas"][c, 0]) <= 1e-12 + 1e-9 * abs(f["support"])
        # and the ADMM iteration never ran beyond the early attempt
        assert g5["admm_iters"][c] == 1


def test_farkas_certificate_rejects_what_is_not_a_ray(otrack):
    N = 10
    lim, w = M.Limits.stock(), M.Weights.stock()
    lb, ub = np.full(N, -0.1), np.full(N, 0.1)
    P, q, A, l, u = M.assemble(otrack, 5, np.array([0.01, 0.05, 0.0]), np.zeros(2 * N), lb, ub, N, w, lim)
    r = O.solve(P, q, A, l, u, O.Settings(polish=2))
    assert r.status == O.SOLVED
    assert not O.farkas_certificate(A, l, u, r.y)["ok"]                  # multipliers of a feasible problem
    assert not O.farkas_certificate(A, l, u, np.zeros(l.size))["ok"]
    rng = np.random.default_rng(0)
    assert not O.farkas_certificate(A, l, u, rng.normal(size=l.size))["ok"]
    lb2, ub2 = lb.copy(), ub.copy()
    lb2[0], ub2[0] = 0.09, 0.1                             # unreachable from e_y = 0.01 in one step
    P, q, A, l, u = M.assemble(otrack, 5, np.array([0.01, 0.0, 0.0]), np.zeros(2 * N), lb2, ub2, N, w, lim)
    r1 = O.solve(P, q, A, l, u, O.Settings(polish=2))                    # phase 1: a handful of interior-point iterations
    r0 = O.solve(P, q, A, l, u, O.Settings(polish=2, phase1=0))          # OSQP's ADMM: hundreds of iterations, same verdict
    assert r1.status == r0.status == O.PRIMAL_INFEASIBLE
    assert r1.iters == 1 and r0.iters > 25
    assert O.farkas_certificate(A, l, u, r1.y, 1e-6)["ok"]
    assert not O.farkas_certificate(A, l, u, -r1.y, 1e-6)["ok"]


@pytest.mark.parametrize("N", [3, 30])
def test_oracle_reproduces_g5(N):
    g4 = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    g5 = np.load(M.GOLDEN + "/g5_solutions_N%d.npz" % N)
    n, m = 5 * N + 3, 8 * N + 6
    for c in range(0, g4["s"].size, 5):
        lo, hi = g4["A_case_ptr"][c], g4["A_case_ptr"][c + 1]
        A = sparse.csc_matrix((g4["A_data"][lo:hi], g4["A_indices"][lo:hi], g4["A_indptr"][c]), shape=(m, n)).toarray()
        r = O.solve(np.diag(g4["P_diag"][c]), g4["q"][c], A, g4["l"][c], g4["u"][c], O.Settings(polish=2))
        assert r.status == g5["status"][c] and r.iters == g5["admm_iters"][c]
        if r.status == 1:
            assert np.max(np.abs(r.x - g5["x"][c])) < 1e-9
