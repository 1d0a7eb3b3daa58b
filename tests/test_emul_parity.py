"""CPU lock-step emulation of the HIP kernels (same source, lane = 64-wide vector) vs the oracle.
This is how the kernel ALGORITHM is checked without a GPU; the -m gpu tests repeat it on hardware."""
import os

import numpy as np
import pytest

import mpc_np as M
import mpmpc
import mpmpc_testlib as T
import osqp_np as O
import scenarios


def _inputs(sc):
    return (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)


@pytest.mark.parametrize("cfgid,B,N", [(2, 12, 30), (4, 12, 30), (3, 6, 50), (2, 6, 10), (4, 6, 3)])
def test_k1_assembly_matches_oracle(cfgid, B, N, emu, track, otrack):
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = T.stock_config(N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    w = M.Weights.time_optimal() if sc.weights == "time_optimal" else M.Weights.stock()
    for i in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], N)
        P0, q0, A0, l0, u0 = M.assemble(otrack, int(sc.wp_id[i]), sc.x0[i], sc.cc_prev[i], sc.lb[i], sc.ub[i], N,
                                        w, M.Limits.stock())
        assert np.array_equal(A, A0) and np.array_equal(Pd, np.diag(P0)) and np.array_equal(q, q0)
        assert np.array_equal(l, l0)
        fin = np.isfinite(u0)
        assert np.array_equal(u[~fin], u0[~fin])
        assert np.max(np.abs(u[fin] - u0[fin])) <= 4 * np.finfo(float).eps      # libm tan, see DESIGN.md
    # table-driven corridor == per-instance corridor rows
    qp_t = emu.assemble(cfg, track, _inputs(sc), use_table=True, obstacles=sc.obstacles)
    assert np.array_equal(qp, qp_t)


def test_k1_matches_reference_capture(emu, track):
    """K1 against what the reference handed to osqp.setup (golden G4, N=30)."""
    g = np.load(M.GOLDEN + "/g4_assembly_N30.npz")
    N = 30
    cfg = T.stock_config(N)
    B = g["s"].size
    qp = emu.assemble(cfg, track, (g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"]))
    for c in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, c, :], N)
        assert np.array_equal(q, g["q"][c]) and np.array_equal(l, g["l"][c]) and np.array_equal(Pd, g["P_diag"][c])
        fin = np.isfinite(g["u"][c])
        assert np.max(np.abs(u[fin] - g["u"][c][fin])) <= 4 * np.finfo(float).eps


@pytest.mark.parametrize("G", [64, 32])
def test_admm_stock_matches_oracle(G, emu, track):
    """OSQP at its defaults (what src/MPC.py:159 runs): same iterates, statuses, iteration counts."""
    sc = scenarios.make(2, track, B=10)
    cfg = T.stock_config(sc.N)
    qp = emu.assemble(cfg, track, _inputs(sc))
    sol = emu.solve(cfg, mpmpc.default_settings(polish=0), qp, G=G)
    for i in range(sc.B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
        r = O.solve(np.diag(Pd), q, A, l, u, O.Settings())
        assert sol.status[i] == r.status and sol.iters[i, 0] == r.iters
        assert np.max(np.abs(sol.z[i] - r.x)) < 1e-7 and np.max(np.abs(sol.y[i] - r.y)) < 1e-7
        assert abs(sol.resid[i, 0] - r.pri_res) < 1e-9 and abs(sol.resid[i, 1] - r.dua_res) < 1e-9


# (G = 64 with N + 1 <= 32 runs the interior-point stage in the split layout: N = 31 fills both half-waves,
#  N = 32 is the first horizon that cannot split)
@pytest.mark.parametrize("cfgid,B,N,G", [(2, 12, 30, 64), (4, 16, 30, 32), (3, 4, 50, 64), (2, 8, 10, 16), (4, 8, 3, 16),
                                         (4, 6, 31, 64), (2, 4, 32, 64), (4, 6, 10, 64), (2, 4, 3, 64)])
def test_certified_matches_oracle(cfgid, B, N, G, emu, track):
    """max |u - u_ref| <= 1e-6 (north-star tolerance); measured ~1e-15."""
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = T.stock_config(N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=G)
    L = scenarios.CAR_LENGTH
    n_cert = n_inf = n_skip = 0
    for i in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], N)
        r = O.solve(np.diag(Pd), q, A, l, u, O.Settings(polish=2))
        if r.polished != 1 and sol.status[i] == 1:
            # the dense oracle interior point gave up where the kernel did not: the kernel's point
            # must then carry its own certificate
            assert O.kkt_certificate(np.diag(Pd), q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
            n_skip += 1
            continue
        assert sol.status[i] == r.status
        # same ADMM iteration count, unless exactly one side certified at the early-polish attempt
        # (the dense numpy interior point is less robust than the kernel's / the C port's)
        assert sol.iters[i, 0] == r.iters or 1 in (sol.iters[i, 0], r.iters)
        n_inf += int(r.status == O.PRIMAL_INFEASIBLE)
        if r.status == O.SOLVED:
            n_cert += 1
            uref = np.array([r.x[3 * (N + 1)], np.arctan(r.x[3 * (N + 1) + 1] * L)])
            assert np.max(np.abs(sol.u0[i] - uref)) <= 1e-6
            e = np.abs(sol.z[i] - r.x)
            e[-1] = 0.0                 # kappa_{N-1} is cost free ...
            e[3 * N + 1] = 0.0          # ... and so is the e_psi_N it alone drives
            assert e.max() <= 1e-6
            assert O.kkt_certificate(np.diag(Pd), q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
    # every instance is accounted for: compared with the oracle's certified optimum, or infeasible on both sides; on these
    # batches the dense numpy oracle never gives up ("weak" 4: the test used to ask for half)
    assert n_skip == 0 and n_cert + n_inf == B, (n_cert, n_inf, n_skip)


def test_ragged_batch_and_group_packing(emu, track):
    """B not a multiple of the instances per wave; G=16/32/64 give the same answers."""
    sc = scenarios.make(2, track, B=7, N=10)
    cfg = T.stock_config(10)
    qp = emu.assemble(cfg, track, _inputs(sc))
    st = mpmpc.default_settings()
    s64, s32, s16 = (emu.solve(cfg, st, qp, G=g) for g in (64, 32, 16))
    for s in (s32, s16):
        assert np.array_equal(s.status, s64.status) and np.array_equal(s.iters, s64.iters)
        assert np.max(np.abs(s.z - s64.z)) < 1e-12


def test_maximum_horizon_and_finite_state_boxes(emu, track):
    """N = 63 (one lane per stage is the limit) with finite boxes on e_psi and t as well."""
    import oracle_c as OC
    N, B = 63, 4
    sc = scenarios.make(2, track, B=B, N=50)                 # corridor tables hold 50 columns
    lb = np.concatenate([sc.lb, np.repeat(sc.lb[:, -1:], N - 50, axis=1)], axis=1)
    ub = np.concatenate([sc.ub, np.repeat(sc.ub[:, -1:], N - 50, axis=1)], axis=1)
    cc = np.zeros((B, 2 * N))
    Q, R, QN = scenarios.WEIGHTS["stock"]
    xmin, xmax = np.array([-np.inf, -0.6, -np.inf]), np.array([np.inf, 0.6, 4.0])
    cfg = mpmpc.make_config(N, Q, R, QN, xmin, xmax, scenarios.UMIN, scenarios.UMAX, 4.0, 0.12)
    qp = emu.assemble(cfg, track, (sc.wp_id, sc.x0, cc, lb, ub))
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    ocfg = OC.mpc_cfg(N, (Q, R, QN), scenarios.UMIN, scenarios.UMAX, xmin, xmax, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, cc, lb, ub)
    assert np.array_equal(sol.status, ref["status"]) and np.all(sol.status == 1)
    assert np.array_equal(sol.iters[:, 0], ref["iters"][:, 0])
    assert np.max(np.abs(sol.u0 - ref["u0"])) <= 1e-6
    assert np.all(np.abs(sol.z[:, 1:3 * (N + 1):3]) <= 0.6 + 1e-9)        # e_psi box respected


def test_c_port_agrees_with_emulation_on_a_larger_sample(emu, track):
    """Statuses, iteration counts and controls over 256 obstacle-corridor instances (incl. infeasible)."""
    import oracle_c as OC
    sc = scenarios.make(4, track, B=256)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    sol = emu.solve(cfg, mpmpc.default_settings(phase1_accept=0), qp, G=64)
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN,
                      scenarios.XMAX, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, sc.cc_prev,
                       sc.lb, sc.ub)
    assert np.array_equal(sol.status, ref["status"])
    assert np.array_equal(sol.iters[:, 0], ref["iters"][:, 0])
    ok = sol.status == 1
    assert ok.sum() > 200 and (sol.status == mpmpc.PRIMAL_INFEASIBLE).sum() > 0
    assert np.max(np.abs(sol.u0[ok] - ref["u0"][ok])) <= 1e-6


def test_non_finite_inputs_get_no_verdict(emu, track):
    """A NaN / Inf pose or previous plan must not come back as a solved plan: status UNSOLVED (-10), which the
    host class treats like an infeasibility verdict (fallback branch); the neighbours in the batch are untouched."""
    sc = scenarios.make(2, track, B=6)
    cfg = T.stock_config(sc.N, sc.weights)
    x0, cc = sc.x0.copy(), sc.cc_prev.copy()
    x0[1, 0], x0[2, 1] = np.nan, np.inf
    cc[4, :] = np.nan
    for G in (64, 32):
        qp = emu.assemble(cfg, track, (sc.wp_id, x0, cc, sc.lb, sc.ub))
        sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=G)
        clean = emu.solve(cfg, mpmpc.default_settings(), emu.assemble(cfg, track, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)), G=G)
        assert list(sol.status[[1, 2]]) == [mpmpc.UNSOLVED, mpmpc.UNSOLVED]
        for i in (0, 3, 5):
            assert sol.status[i] == clean.status[i] == 1 and np.max(np.abs(sol.u0[i] - clean.u0[i])) <= 1e-9
        assert sol.status[4] in (1, mpmpc.UNSOLVED)        # a NaN previous plan only disables the speed cap or poisons it


@pytest.mark.parametrize("cfgid,G", [(2, 64), (4, 32), (3, 64)])
def test_warm_start_reproduces_the_cold_solution(cfgid, G, emu, track):
    """Closed-loop warm start: active-set rounds from a given active set.  From the cold solve's own active set
    every certified instance is certified again without a single ADMM / interior-point iteration and with the
    same plan; from a WRONG guess (all bounds active) the normal path takes over and the answer is the same."""
    sc = scenarios.make(cfgid, track, B=12)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc), obstacles=sc.obstacles)
    st = mpmpc.default_settings()
    cold, act = emu.solve_warm(cfg, st, qp, np.zeros((sc.B, mpmpc.stage_ld(sc.N)), np.int32), G=G)
    ref, _ = emu.solve_launch(cfg, st, qp, G=G)          # the launcher's cold sequence of kernels
    ok = cold.status == 1
    # no guess: the plain path (the closed loop hands its tail to the general kernel, the batch launch to the tail solver:
    # an infeasible instance's least-violation point agrees to ~1e-8 between the two)
    assert np.array_equal(cold.status, ref.status) and np.array_equal(cold.u0[ok], ref.u0[ok]) and np.max(np.abs(cold.u0 - ref.u0)) <= 1e-6
    assert np.all((act[ok, 0] >> 30) & 1) and not np.any(act[~ok])
    warm, act2 = emu.solve_warm(cfg, st, qp, act, G=G)
    assert np.array_equal(warm.status, cold.status)
    assert np.all(warm.iters[ok] == 0)
    assert np.max(np.abs(warm.u0[ok] - cold.u0[ok])) <= 1e-9 and np.array_equal(act2[ok], act[ok])
    wrong = np.where(act != 0, (1 << 30) | 0x3FF, 0).astype(np.int32)
    again, _ = emu.solve_warm(cfg, st, qp, wrong, G=G)
    assert np.array_equal(again.status, cold.status)
    assert np.max(np.abs(again.u0[ok] - cold.u0[ok])) <= 1e-9


@pytest.mark.parametrize("N,G", [(3, 64), (3, 16), (10, 64), (10, 16), (30, 64), (30, 32), (50, 64)])
def test_emulated_kernels_reach_the_g5_optima_of_the_reference_qps(N, G, emu, track):
    """The reference's own captured inputs (G4: wp_id, x0, previous plan, corridor rows) through K1 + K2 as emulated
    lane code: statuses of G5 and its certified optima to 1e-6 (the north-star tolerance; measured ~1e-8)."""
    g4 = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    g5 = np.load(M.GOLDEN + "/g5_solutions_N%d.npz" % N)
    cfg = T.stock_config(N, str(g4["weights"][0]))
    qp = emu.assemble(cfg, track, (g4["wp_id"].astype(np.int32), g4["x0"], g4["cc_prev"], g4["lb"], g4["ub"]))
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=G)
    assert np.array_equal(sol.status, g5["status"])
    ok = g5["status"] == 1
    assert np.max(np.abs(sol.z[ok] - g5["x"][ok])) < 1e-6
    assert np.max(np.abs(sol.z[ok][:, -2 * N:-2 * N + 2] - g5["x"][ok][:, -2 * N:-2 * N + 2])) < 1e-8      # (v_0, kappa_0)


# ---------------------------------------------------------------------------------------------------------------
# Phase 1: infeasible instances leave with a Farkas ray after a few interior-point iterations
# ---------------------------------------------------------------------------------------------------------------
def _farkas_ok(qp_i, N, y, eps=1e-6):
    """OSQP's primal-infeasibility criterion on the dense (A, l, u) rebuilt from K1's fields: plain numpy."""
    Pd, q, A, l, u = T.qp_to_dense(qp_i, N)
    return O.farkas_certificate(A, l, u, y, eps)["ok"]


@pytest.mark.parametrize("G", [64, 32])
def test_phase1_certifies_every_infeasible_instance_without_admm(G, emu, track):
    """512 obstacle-corridor instances (config 4): every instance ends as a certified optimum or with a certified
    Farkas ray, nobody runs more than the ONE ADMM iteration of the early attempt, statuses and iteration counts are
    the C oracle's, and the rays check out with plain numpy on K1's output."""
    import oracle_c as OC
    sc = scenarios.make(4, track, B=512)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    sol, n_tail = emu.solve_launch(cfg, mpmpc.default_settings(phase1_accept=0), qp, G=G)
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN,
                      scenarios.XMAX, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, sc.cc_prev,
                       sc.lb, sc.ub, want_y=True)
    assert np.array_equal(sol.status, ref["status"])
    assert set(np.unique(sol.status)) == {1, mpmpc.PRIMAL_INFEASIBLE}
    assert np.all(sol.iters[:, 0] == 1) and np.all(ref["iters"][:, 0] == 1)
    inf = np.flatnonzero(sol.status == mpmpc.PRIMAL_INFEASIBLE)
    assert inf.size >= 30
    # the reduced-native launch (any packing) hands exactly those to the general kernel; with native = 0 a packed launch does
    assert n_tail == inf.size
    for i in inf:
        assert _farkas_ok(qp[:, i, :], sc.N, sol.y[i]) and _farkas_ok(qp[:, i, :], sc.N, ref["y"][i])
        assert sol.resid[i, 0] > 1e-4                      # the least-violation point does violate a bound
    ok = sol.status == 1
    assert np.max(np.abs(sol.u0[ok] - ref["u0"][ok])) <= 1e-6
    # interior-point iterations of an infeasible instance: failed attempt (stopped by the divergence test) + phase 1
    assert np.median(sol.iters[inf, 1]) <= 14


def test_phase1_off_restores_osqps_own_verdicts(emu, track):
    """phase1 = 0: what the early attempt cannot certify runs OSQP's ADMM to its own verdict (hundreds of iterations),
    as in round 1; the infeasible set it finds is a subset of the one phase 1 proves (ADMM at eps = 1e-3 calls marginally
    infeasible instances "solved", and the polish then flags them inaccurate)."""
    sc = scenarios.make(4, track, B=96)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    on = emu.solve(cfg, mpmpc.default_settings(phase1_accept=0), qp, G=64)
    off = emu.solve(cfg, mpmpc.default_settings(phase1=0), qp, G=64)
    solved = on.status == 1
    assert np.array_equal(off.status[solved], on.status[solved]) and np.array_equal(off.u0[solved], on.u0[solved])
    rest = ~solved
    assert rest.sum() >= 5 and np.all(on.status[rest] == mpmpc.PRIMAL_INFEASIBLE)
    assert np.all(np.isin(off.status[rest], (mpmpc.PRIMAL_INFEASIBLE, mpmpc.SOLVED_INACCURATE, mpmpc.MAX_ITER_REACHED)))
    assert np.all(off.iters[rest, 0] >= 25) and np.all(on.iters[rest, 0] == 1)
    for i in np.flatnonzero(rest):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
        r = O.solve(np.diag(Pd), q, A, l, u, O.Settings(polish=2, phase1=0))
        assert r.status == off.status[i] and r.iters == off.iters[i, 0]


def test_phase1_leaves_feasible_batches_alone(emu, track):
    """Configs 2 and 3 never reach phase 1: identical outputs with phase1 on and off."""
    for cid, B in ((2, 64), (3, 32)):
        sc = scenarios.make(cid, track, B=B)
        cfg = T.stock_config(sc.N, sc.weights)
        qp = emu.assemble(cfg, track, _inputs(sc))
        a = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
        b = emu.solve(cfg, mpmpc.default_settings(phase1=0), qp, G=64)
        assert np.all(a.status == 1) and np.array_equal(a.z, b.z) and np.array_equal(a.iters, b.iters)


def test_second_polish_attempt_from_phase1s_point(emu, track):
    """tests/golden/p1_retry_N3.npy: the assembled stage fields of ONE feasible N = 3 instance of config 4 (found by
    profiles/stress.py, seed 1, trial 13, instance 16) on which the WARM-started interior point (ipm_start_mu = 0: start
    from the multipliers of the one ADMM iteration) jams next to a degenerate vertex.  Phase 1 finds it feasible; the
    second attempt from phase 1's point certifies the optimum with no ADMM iteration beyond the first.  Without phase 1
    the instance goes the long way.  The default, centred start does not jam on it in the first place."""
    qp = np.load(os.path.join(os.path.dirname(__file__), "golden", "p1_retry_N3.npy"))
    cfg = T.stock_config(3, scenarios.CONFIGS[4]["weights"])
    Pd, q, A, l, u = T.qp_to_dense(qp[:, 0, :], 3)
    ref = O.solve(np.diag(Pd), q, A, l, u, O.Settings(polish=2, ipm_start_mu=0.0))
    assert ref.status == 1 and ref.iters == 1       # (the oracle's full-problem iteration does not jam on it)
    for G in (64, 32, 16):
        s = emu.solve(cfg, mpmpc.default_settings(ipm_start_mu=0.0), qp, G=G)
        if G == 64:
            assert s.status[0] == 1 and s.iters[0, 0] == 1 and s.iters[0, 1] > 30      # 30 jammed, phase 1, retry
            np.testing.assert_allclose(s.z[0], ref.x, atol=1e-7)
        else:       # packed kernels carry no phase 1: the instance is handed to the tail launch (status 0 here)
            assert s.status[0] in (0, 1)
    off = emu.solve(cfg, mpmpc.default_settings(phase1=0, ipm_start_mu=0.0), qp, G=64)
    assert off.iters[0, 0] > 25
    dflt = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    assert dflt.status[0] == 1 and dflt.iters[0, 0] == 1 and dflt.iters[0, 1] < 15
    np.testing.assert_allclose(dflt.z[0], ref.x, atol=1e-7)


def test_controls_vs_reference_counts_only_certified_equal_objective_points_as_alternatives(emu, track):
    """The comparison helper of the full-batch parity tests: identical answers -> nothing to report; a reference whose
    control is off and whose plan is NOT an optimum (objective differs, no certificate) is a disagreement, not an
    "alternative optimum"."""
    sc = scenarios.make(4, track, B=48)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    ref = dict(status=sol.status.copy(), u0=sol.u0.copy(), z=sol.z.copy(), y=sol.y.copy())
    worst, alt = T.controls_vs_reference(qp, sc.N, sol, ref, 1e-6)
    assert worst == 0.0 and alt.size == 0
    i = int(np.flatnonzero(sol.status == 1)[0])
    ref["u0"][i, 1] += 1e-3
    ref["z"][i, 3 * (sc.N + 1) + 1] += 1e-3            # kappa_0 of the reference plan moved: no longer a KKT point
    worst, alt = T.controls_vs_reference(qp, sc.N, sol, ref, 1e-6)
    assert alt.size == 0 and worst >= 0.9e-3


def test_default_branch_agreement_with_restated_stock_osqp_in_the_emulation(emu, track):
    """The CPU twin of tests/test_gpu_parity.py::test_default_branch_agreement_... on the FULL config 4 (8 192 instances): the
    launcher's own sequence of kernels, emulated, on every instance the certified C port proves infeasible (the others are
    "solved" on both sides), against the restated stock OSQP.  At most 8 disagreements, all "device refuses, OSQP returns a
    plan", all abandoned by OSQP at max_iter or within 0.5 % of its primal tolerance; and the least violation phase 1 reports
    IS the primal residual OSQP's ADMM iteration converges to (to 1 %) where that iteration ran into max_iter."""
    import oracle_c as OC
    sc = scenarios.make(4, track, B=8192)
    cfg = T.stock_config(sc.N, sc.weights)
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN, scenarios.XMAX, 4.0, 0.12)
    args = (track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    stock = OC.mpc_batch(ocfg, OC.settings(polish=0, early_polish=0, phase1=0), *args)
    cert = OC.mpc_batch(ocfg, OC.settings(), *args)
    inf = np.flatnonzero(cert["status"] == -3)
    assert inf.size > 600 and np.all(np.isin(stock["status"][cert["status"] == 1], (1, 2)))
    qp = emu.assemble(cfg, track, (sc.wp_id[inf], sc.x0[inf], sc.cc_prev[inf], sc.lb[inf], sc.ub[inf]))
    sol, _ = emu.solve_launch(cfg, mpmpc.default_settings(), qp, G=32)
    assert set(np.unique(sol.status)) <= {2, -3} and (sol.status == 2).sum() >= 60
    dev_usable, st_usable = sol.status == 2, np.isin(stock["status"][inf], (1, 2, -2))
    dis = np.flatnonzero(dev_usable != st_usable)
    thr = 1e-3 + 1e-3 * scenarios.UMAX[1]
    assert dis.size <= 8
    for j in dis:
        assert not dev_usable[j] and (stock["iters"][inf[j], 0] >= 4000 or abs(sol.resid[j, 0] / thr - 1.0) <= 5e-3), (inf[j], sol.resid[j, 0])
    cap = stock["iters"][inf, 0] >= 4000
    assert cap.sum() >= 3 and np.max(np.abs(sol.resid[cap, 0] / stock["resid"][inf[cap], 0] - 1.0)) <= 1e-2


# ---------------------------------------------------------------------------------------------------------------
# full terminal weight QN (src/MPC.py:150,154 use the whole matrix)
# ---------------------------------------------------------------------------------------------------------------
QN_FULL = scenarios.QN_FULL


def _dense_with_qn(qp_i, N, QN):
    Pd, q, A, l, u = T.qp_to_dense(qp_i, N)
    P = np.diag(Pd)
    P[3 * N:3 * N + 3, 3 * N:3 * N + 3] = QN
    return P, q, A, l, u


@pytest.mark.parametrize("cfgid,N,B", [(2, 30, 12), (4, 30, 16), (3, 50, 6), (2, 10, 8)])
def test_full_terminal_weight_matches_oracle(cfgid, N, B, emu, track, otrack):
    """A QN with off-diagonal entries: K1's cost vector is the reference's -QN.xr (checked against the numpy restatement
    of src/MPC.py:150-155 with the full matrix), and K2 reaches the oracle's certified optimum of the QP with the dense
    terminal block - statuses, z to 1e-6, KKT certificate on the dense data."""
    sc = scenarios.make(cfgid, track, B=B, N=N)
    Q, R, _ = scenarios.WEIGHTS[sc.weights]
    cfg = mpmpc.make_config(N, Q, R, QN_FULL, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                            scenarios.AY_MAX, scenarios.CAR_LENGTH)
    assert list(cfg.QN) == [1.0, 0.5, 0.4] and list(cfg.QN_offdiag) == [0.3, -0.1, 0.2]
    qp = emu.assemble(cfg, track, _inputs(sc))
    w = M.Weights(np.diag(Q), np.diag(R), QN_FULL)
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    n_ok = 0
    for i in range(B):
        P, q, A, l, u = _dense_with_qn(qp[:, i, :], N, QN_FULL)
        P0, q0, A0, l0, u0 = M.assemble(otrack, int(sc.wp_id[i]), sc.x0[i], sc.cc_prev[i], sc.lb[i], sc.ub[i], N, w,
                                        M.Limits.stock())
        assert np.array_equal(P, P0) and np.array_equal(q, q0) and np.array_equal(A, A0)
        r = O.solve(P, q, A, l, u, O.Settings(polish=2))
        if r.polished != 1 and sol.status[i] == 1:
            # (the dense numpy interior point is less robust than the kernel's at N = 50: the device's point then has
            #  to stand on the solver-independent certificate alone)
            assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
            continue
        assert sol.status[i] == r.status, (i, sol.status[i], r.status)
        if r.status == 1:
            n_ok += 1
            e = np.abs(sol.z[i] - r.x)
            e[-1] = 0.0
            assert e.max() <= 1e-6
            assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
    assert n_ok >= B // 2
    # and with the off-diagonals set to zero the FQ code path reproduces the diagonal kernels bit for bit
    with pytest.raises(ValueError):
        mpmpc.make_config(N, Q, R, np.array([[1.0, 0.2, 0], [0.1, 1, 0], [0, 0, 1]]), scenarios.XMIN, scenarios.XMAX,
                          scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH)


# ---------------------------------------------------------------------------------------------------------------
# full stage weights: Q and R with off-diagonal entries (src/MPC.py:150 puts the whole matrices into P, src/MPC.py:153-155
# only their diagonals into q)
# ---------------------------------------------------------------------------------------------------------------
# (the weight sets live beside the workload generator: scenarios.py - positive definite with every off-diagonal set, and
#  positive SEMI-definite, singular blocks: Q of rank one (no cost on t at all), R of rank one)
Q_FULL, R_FULL, Q_RANK1, R_RANK1 = scenarios.Q_FULL, scenarios.R_FULL, scenarios.Q_RANK1, scenarios.R_RANK1
FULL_WEIGHT_SETS = scenarios.FULL_WEIGHT_SETS


def full_weight_config(N, name, max_batch=1):
    Q, R, QN = FULL_WEIGHT_SETS[name]
    return mpmpc.make_config(N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX,
                             scenarios.CAR_LENGTH, max_batch=max_batch)


@pytest.mark.parametrize("name,cfgid,N,B", [("full", 2, 30, 12), ("full", 4, 30, 16), ("full", 3, 50, 6), ("full", 2, 10, 8), ("full", 4, 3, 8),
                                            ("q_only", 4, 30, 8), ("r_only", 4, 30, 8), ("rank1", 2, 30, 8), ("rank1", 4, 10, 8)])
def test_full_stage_weights_match_oracle(name, cfgid, N, B, emu, track, otrack):
    """Q, R (and QN) with off-diagonal entries - the input class the reference accepts and rounds 1 - 4 refused (VERDICT r4
    item 1).  K1's fields + the configuration's off-diagonals ARE the reference's (P, q): bit-equal with the numpy restatement
    of src/MPC.py:150-155 for the whole matrices (q from the diagonals only - the reference's quirk).  K2 (general kernel,
    dense 3 x 3 / 2 x 2 stage blocks) reaches the dense oracle's certified optimum: statuses, z to 1e-6, the plain-numpy KKT
    certificate on the dense data."""
    Q, R, QN = FULL_WEIGHT_SETS[name]
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = full_weight_config(N, name)
    assert list(cfg.Q) == list(np.diag(Q)) and list(cfg.Q_offdiag) == [Q[0, 1], Q[0, 2], Q[1, 2]] and list(cfg.R_offdiag) == [R[0, 1]]
    qp = emu.assemble(cfg, track, _inputs(sc))
    w = M.Weights(Q, R, QN)
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    n_ok = 0
    for i in range(B):
        P, q, A, l, u = T.qp_to_dense_full(qp[:, i, :], N, cfg)
        P0, q0, A0, l0, u0 = M.assemble(otrack, int(sc.wp_id[i]), sc.x0[i], sc.cc_prev[i], sc.lb[i], sc.ub[i], N, w, M.Limits.stock())
        assert np.array_equal(P, P0) and np.array_equal(q, q0) and np.array_equal(A, A0)
        r = O.solve(P, q, A, l, u, O.Settings(polish=2))
        if r.polished != 1 and sol.status[i] == 1:
            assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
            continue
        assert sol.status[i] == r.status or (sol.status[i] == 2 and r.status == -3), (i, sol.status[i], r.status)
        if r.status == 1:
            n_ok += 1
            cert = O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])
            assert cert["ok_tol"](1e-8), (i, cert)
            # (singular weight blocks leave flat directions: compare where the optimum is unique - the objective always is)
            f = lambda x: 0.5 * x @ P @ x + q @ x
            assert abs(f(sol.z[i]) - f(r.x)) <= 1e-9 * max(1.0, abs(f(r.x)))
            if name != "rank1":
                e = np.abs(sol.z[i] - r.x)
                e[-1] = 0.0
                assert e.max() <= 1e-6, (i, e.max())
    assert n_ok >= B // 2
    with pytest.raises(ValueError):
        mpmpc.make_config(N, np.array([[1.0, 0.2, 0], [0.1, 1, 0], [0, 0, 1]]), R, QN, scenarios.XMIN, scenarios.XMAX,
                          scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH)


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_full_stage_weights_on_the_references_own_captures(N, emu, track):
    """Golden G4f: what the REFERENCE handed to osqp.setup with non-diagonal Q, R, QN.  K1's fields plus the configuration's
    off-diagonals rebuild its (P, q, l) bit for bit (u to the ulps of tan), and K2 - general kernel, dense stage blocks -
    reaches the dense oracle's certified optimum of the CAPTURED QP: same statuses, z to 1e-6, first control to 1e-8, KKT
    certificate with plain numpy on the captured data."""
    from scipy import sparse
    g = np.load(M.GOLDEN + "/g4f_assembly_N%d.npz" % N)
    cfg = full_weight_config(N, "full")
    B, n, m = g["s"].size, 5 * N + 3, 8 * N + 6
    qp = emu.assemble(cfg, track, (g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"]))
    sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    n_ok = 0
    for c in range(B):
        P, q, A, l, u = T.qp_to_dense_full(qp[:, c, :], N, cfg)
        Pref = sparse.coo_matrix((g["P_val"][c], (g["P_row"][c], g["P_col"][c])), shape=(n, n)).toarray()
        lo, hi = g["A_case_ptr"][c], g["A_case_ptr"][c + 1]
        Aref = sparse.csc_matrix((g["A_data"][lo:hi], g["A_indices"][lo:hi], g["A_indptr"][c]), shape=(m, n)).toarray()
        assert np.array_equal(P, Pref) and np.array_equal(q, g["q"][c]) and np.array_equal(A, Aref)
        lref = np.where(g["l"][c] <= -1e30, -np.inf, g["l"][c])
        assert np.array_equal(l, lref)
        fin = np.isfinite(g["u"][c])
        assert np.max(np.abs(u[fin] - g["u"][c][fin])) <= 4 * np.finfo(float).eps
        r = O.solve(Pref, g["q"][c], Aref, g["l"][c], g["u"][c], O.Settings(polish=2))
        if r.polished != 1 and sol.status[c] == 1:
            assert O.kkt_certificate(Pref, g["q"][c], Aref, g["l"][c], g["u"][c], sol.z[c], sol.y[c])["ok_tol"](1e-8)
            continue
        assert sol.status[c] == r.status or (sol.status[c] == 2 and r.status == -3), (c, sol.status[c], r.status)
        if r.status == 1:
            n_ok += 1
            assert O.kkt_certificate(Pref, g["q"][c], Aref, g["l"][c], g["u"][c], sol.z[c], sol.y[c])["ok_tol"](1e-8)
            e = np.abs(sol.z[c] - r.x)
            e[-1] = 0.0
            assert e.max() <= 1e-6 and np.max(e[3 * (N + 1):3 * (N + 1) + 2]) <= 1e-8
    assert n_ok >= B // 2


def test_empty_speed_box_is_reported_infeasible(emu, track):
    """umin[0] above the curvature-dependent speed cap (src/MPC.py:111-113): an empty interval row.  Stock OSQP refuses
    such data at setup; the build reports the instance infeasible (zero ray, the gap as violation) - device code and
    oracle alike - and leaves the other instances of the batch alone."""
    sc = scenarios.make(2, track, B=8)
    Q, R, QN = scenarios.WEIGHTS["stock"]
    umin = scenarios.UMIN.copy()
    umin[0] = 0.9
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, umin, scenarios.UMAX, scenarios.AY_MAX,
                            scenarios.CAR_LENGTH)
    cc = sc.cc_prev.copy()
    cc[:4, 1::2], cc[:4, 0::2] = 0.6, 1.0            # large predicted steering: speed cap 0.12 < umin 0.9
    cc[4:] = 0.0                                     # cold plan: no cap
    qp = emu.assemble(cfg, track, (sc.wp_id, sc.x0, cc, sc.lb, sc.ub))
    assert qp[15, :4, :sc.N].min() < 0.9 and qp[15, 4:, :sc.N].min() >= 0.9
    for G in (64, 32):
        sol = emu.solve(cfg, mpmpc.default_settings(), qp, G=G)
        assert np.all(sol.status[:4] == mpmpc.PRIMAL_INFEASIBLE) and np.all(sol.y[:4] == 0.0)
        assert np.all(sol.resid[:4, 0] > 0.5)
        assert np.all(sol.status[4:] == 1)
    Pd, q, A, l, u = T.qp_to_dense(qp[:, 0, :], sc.N)
    r = O.solve(np.diag(Pd), q, A, l, u, O.Settings(polish=2))
    assert r.status == O.PRIMAL_INFEASIBLE and r.pri_res > 0.5


@pytest.mark.parametrize("cfgid,B,N,Gs", [(2, 96, 30, (64, 32)), (4, 128, 30, (64, 32)), (2, 48, 10, (64, 32, 16)), (4, 48, 3, (64, 32, 16)),
                                          (2, 48, 15, (64, 32, 16)), (4, 64, 20, (64, 32)), (2, 32, 31, (64, 32)), (4, 32, 16, (64, 32)),
                                          (4, 32, 7, (64, 32, 16))])
def test_cyclic_reduction_factorisation_agrees_with_the_sequential_one(cfgid, B, N, Gs, emu, track):
    """The reduced-native kernels factor the chains of the Schur complement by cyclic reduction in Cholesky form
    (mpmpc_core.hpp, factor_cr2 / s_solve_cr2: a different elimination ORDER of the same SPD block-tridiagonal matrix).
    Against the same kernel with the chain-sequential elimination (CR = false): same verdicts, same hand-overs to the tail,
    the same interior-point iteration counts instance by instance, the same controls to rounding - and the points pass the
    plain-numpy KKT test.  (All horizons 3 .. 31 on both configurations were swept once with the same checks: no difference.)"""
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = T.stock_config(N, sc.weights)
    st = mpmpc.default_settings()
    qp = emu.assemble(cfg, track, _inputs(sc))
    for G in Gs:
        a, ta = emu.solve_rn(cfg, st, qp, G)
        b, tb = emu.solve_rn(cfg, st, qp, G, sequential=True)
        assert np.array_equal(a.status, b.status) and ta == tb
        ok = a.status == 1
        assert ok.sum() >= B // 2
        assert np.array_equal(a.iters[ok], b.iters[ok])
        assert np.max(np.abs(a.u0[ok] - b.u0[ok])) <= 1e-13
        assert np.max(np.abs(a.z[ok] - b.z[ok])) <= 1e-11
        prim, stat, comp = T.kkt_batch(qp[:, ok], N, a.z[ok], a.y[ok])
        assert max(prim.max(), stat.max(), comp.max()) <= 1e-9


# ---- round 4: the terminal-time kernels (csrc/mpmpc_reduced_t.hpp) and the cyclic reduction of the 32-lane chains
@pytest.mark.parametrize("cfgid,N,B", [(3, 50, 24), (3, 30, 24), (4, 50, 64), (4, 30, 64), (3, 10, 12), (3, 3, 12), (3, 33, 8), (3, 47, 8)])
def test_terminal_time_kernel_against_the_general_kernel_and_the_oracle(cfgid, N, B, emu, track):
    """Weightings with a terminal cost on the time state (QN[2] > 0 = Q[2]: BASELINE config 3) run the reduced-native kernel
    of mpmpc_reduced_t.hpp - t eliminated, the (e_y, e_psi, kappa, v) QP with one rank-one term, Sherman-Morrison on the
    2 x 2-block solves - and what it cannot certify goes to the general kernel's tail launch.  The launcher's sequence against
    the general 3-state kernel alone (statuses, plan to 1e-8), against the FULL problem's KKT system in plain numpy, against
    Farkas' lemma for the refusals, and against the dense oracle; cfgid 4 = the obstacle corridor (infeasible and marginal
    instances: the tail path) with the time-optimal weights."""
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = T.stock_config(N, "time_optimal")
    st = mpmpc.default_settings()
    assert emu.lib.emu_reduced_native_tt(__import__("ctypes").byref(cfg), __import__("ctypes").byref(st)) == 1
    qp = emu.assemble(cfg, track, _inputs(sc))
    gen = emu.solve(cfg, st, qp, G=64)                     # the general kernel alone
    sol, n_tail = emu.solve_launch(cfg, st, qp, G=64)      # terminal-time kernel + tail launch
    assert np.array_equal(sol.status, gen.status)
    ok = sol.status == 1
    assert ok.sum() >= 0.75 * B and (cfgid != 4 or (n_tail >= 1 and (sol.status == mpmpc.PRIMAL_INFEASIBLE).any()))
    e = np.abs(sol.z - gen.z)
    e[:, -1] = 0.0                                         # kappa_{N-1} is cost free, and so is the e_psi_N it alone drives
    e[:, 3 * N + 1] = 0.0
    assert e[ok].max() <= 1e-8 and np.abs(sol.u0 - gen.u0)[ok].max() <= 1e-9
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, sol.z[ok], sol.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    # the time rows' multipliers are all w t_N (stationarity in t_k): the rank-one term of the eliminated problem
    ne = 3 * (N + 1)
    nu_t = sol.y[ok][:, 2:ne:3]
    assert np.max(np.abs(nu_t - (cfg.QN[2] * sol.z[ok][:, ne - 1])[:, None])) <= 1e-12
    inf = sol.status == mpmpc.PRIMAL_INFEASIBLE
    if inf.any():
        okf, support, aty = T.farkas_batch(qp[:, inf, :], N, sol.y[inf])
        assert okf.all(), (support.max(), aty.max())
    for i in np.flatnonzero(ok)[:6]:
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], N)
        r = O.solve(np.diag(Pd), q, A, l, u, O.Settings(polish=2))
        if r.status == O.SOLVED and r.polished == 1:
            uref = np.array([r.x[3 * (N + 1)], np.arctan(r.x[3 * (N + 1) + 1] * scenarios.CAR_LENGTH)])
            assert np.max(np.abs(sol.u0[i] - uref)) <= 1e-6


@pytest.mark.parametrize("cfgid,N,weights", [(3, 50, "time_optimal"), (3, 40, "time_optimal"), (2, 50, "stock"), (4, 33, "stock"), (2, 45, "stock")])
def test_cyclic_reduction_of_the_32_lane_chains_agrees_with_the_sequential_elimination(cfgid, N, weights, emu, track):
    """N + 1 > 32: a chain of the factorisation is TWO rows of 16 lanes (kCR32: in-row levels, then the first row's survivor
    against the second row's, then the junction).  Same kernel with the chain-sequential elimination (CR = false): identical
    statuses and interior-point iteration counts, the plan to 1e-9 - for the terminal-time kernels and the stock-weight ones."""
    sc = scenarios.make(cfgid, track, B=40, N=N)
    cfg = T.stock_config(N, weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    a, ta = emu.solve_rn(cfg, mpmpc.default_settings(), qp, G=64)
    b, tb = emu.solve_rn(cfg, mpmpc.default_settings(), qp, G=64, sequential=True)
    assert ta == tb and np.array_equal(a.status, b.status) and np.array_equal(a.iters, b.iters)
    ok = a.status == 1
    assert ok.sum() >= 30 and np.abs(a.z[ok] - b.z[ok]).max() <= 1e-9 and np.abs(a.u0[ok] - b.u0[ok]).max() <= 1e-10


@pytest.mark.parametrize("cfgid,B,N,accept", [(4, 768, 30, 1), (4, 384, 30, 0), (5, 512, 30, 1), (4, 192, 10, 1), (4, 96, 3, 1)])
def test_reduced_native_tail_solver_gives_the_general_kernels_answers(cfgid, B, N, accept, emu, track):
    """The tail of a reduced-native launch goes to ReducedTailSolver (mpmpc_reduced_tail.hpp: phase 1 and one more attempt of
    the certified polish, three entries per lane) before the general kernel sees it.  Its one-instance-per-wave form
    (emu_set_lean_tail(2): the split layout, the general kernel's own phase-1 routine) against the same launch with the general
    kernel on the whole tail (emu_set_lean_tail(0), the sequence of rounds 2 - 3): statuses identical; points, multipliers and
    residuals equal to rounding - the two run the same interior point on the same scaling; and the tail solver leaves the general
    kernel nothing on these batches (infeasible, marginally infeasible and capped instances are all it gets).  (The default,
    two instances per wave, is compared with this form in test_two_tail_instances_per_wave_give_the_same_verdicts.)"""
    import ctypes as C
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    st = mpmpc.default_settings(phase1_accept=accept)
    assert emu.lib.emu_reduced_native_tail(C.byref(cfg), C.byref(st)) == 1
    try:
        emu.lib.emu_set_lean_tail(0)
        gen, n_tail = emu.solve_launch(cfg, st, qp, G=32 if N + 1 <= 32 else 64)
        assert emu.lib.emu_last_tail2() == n_tail            # (knob off: the general kernel gets the whole tail)
        emu.lib.emu_set_lean_tail(2)
        lean, n_tail2 = emu.solve_launch(cfg, st, qp, G=32 if N + 1 <= 32 else 64)
        left = emu.lib.emu_last_tail2()
    finally:
        emu.lib.emu_set_lean_tail(1)
    assert n_tail2 == n_tail and n_tail >= 3
    assert left == 0
    assert np.array_equal(lean.status, gen.status)
    assert np.count_nonzero(lean.status == mpmpc.PRIMAL_INFEASIBLE) >= 2
    if accept and N == 30:
        assert np.count_nonzero(lean.status == mpmpc.SOLVED_INACCURATE) >= 2
    np.testing.assert_allclose(lean.z, gen.z, rtol=0, atol=1e-12)
    np.testing.assert_allclose(lean.u0, gen.u0, rtol=0, atol=1e-12)
    scale = np.maximum(1.0, np.abs(gen.y).max(axis=1, keepdims=True))
    assert np.max(np.abs(lean.y - gen.y) / scale) <= 1e-12
    np.testing.assert_allclose(lean.resid, gen.resid, rtol=1e-9, atol=1e-15)
    assert np.array_equal(lean.iters[:, 0], gen.iters[:, 0])
    # (interior-point iterations: the attempt after phase 1 runs ReducedSolver's tolerance ladder, not Solver::polish's)
    assert np.max(np.abs(lean.iters[:, 1] - gen.iters[:, 1])) <= 4


def test_where_the_reduced_native_tail_solver_applies(emu):
    """Every configuration the reduced-native first kernel runs (stock-type weights, any horizon up to 63) with phase 1 on."""
    import ctypes as C
    st = mpmpc.default_settings()
    for N, want in ((3, 1), (30, 1), (31, 1), (32, 1), (50, 1)):
        cfg = T.stock_config(N, "stock")
        assert emu.lib.emu_reduced_native_tail(C.byref(cfg), C.byref(st)) == want, N
    cfg = T.stock_config(30, "time_optimal")           # the terminal-time weights have their own kernel, and the general tail
    assert emu.lib.emu_reduced_native_tail(C.byref(cfg), C.byref(st)) == 0
    off = mpmpc.default_settings(phase1=0)
    assert emu.lib.emu_reduced_native_tail(C.byref(T.stock_config(30, "stock")), C.byref(off)) == 0


def test_reduced_native_tail_solver_fallbacks(emu, track):
    """The two ways out of ReducedTailSolver that the BASELINE batches do not take, forced with as_rounds = 0 (no active-set
    round: no attempt can certify anything).  A marginally infeasible instance then ends with phase 1's least-violation point
    itself - status SOLVED_INACCURATE, zero multipliers, the violation in resid[0]; a feasible instance stays UNSOLVED in the
    tail solver and goes on to the general kernel (full OSQP run).  Both exactly as with the general kernel on the whole tail."""
    sc = scenarios.make(4, track, B=512)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    base, _ = emu.solve_launch(cfg, mpmpc.default_settings(), qp, G=32)
    pick = np.concatenate([np.flatnonzero(base.status == 2)[:3], np.flatnonzero(base.status == mpmpc.PRIMAL_INFEASIBLE)[:2],
                           np.flatnonzero(base.status == 1)[:3]])
    assert pick.size == 8
    qps = np.ascontiguousarray(qp[:, pick, :])
    st = mpmpc.default_settings(as_rounds=0)
    try:
        emu.lib.emu_set_lean_tail(0)
        gen, n_tail = emu.solve_launch(cfg, st, qps, G=32)
        emu.lib.emu_set_lean_tail(2)
        lean, n_tail2 = emu.solve_launch(cfg, st, qps, G=32)
        left = emu.lib.emu_last_tail2()
    finally:
        emu.lib.emu_set_lean_tail(1)
    assert n_tail == n_tail2 == 8
    assert left == 3                                       # the three feasible instances
    # (the default form, two instances per wave: the same verdicts, the same three left over)
    packed, _ = emu.solve_launch(cfg, st, qps, G=32)
    assert emu.lib.emu_last_tail2() == 3 and np.array_equal(packed.status, gen.status) and np.array_equal(packed.iters, gen.iters)
    np.testing.assert_allclose(packed.z, gen.z, rtol=0, atol=1e-6)
    assert np.all(packed.y[:3] == 0.0)
    assert list(lean.status) == [2, 2, 2, mpmpc.PRIMAL_INFEASIBLE, mpmpc.PRIMAL_INFEASIBLE, 2, 2, 2] == list(gen.status)
    assert np.all(lean.iters[5:, 0] > 25) and np.all(lean.iters[:5, 0] == 1)
    assert np.array_equal(lean.iters, gen.iters)
    np.testing.assert_allclose(lean.z, gen.z, rtol=0, atol=1e-12)
    np.testing.assert_allclose(lean.y, gen.y, rtol=0, atol=1e-12)
    np.testing.assert_allclose(lean.resid, gen.resid, rtol=1e-12, atol=0)
    assert np.all(lean.y[:3] == 0.0)                       # a bare least-violation point carries no multipliers
    assert np.all((lean.resid[:3, 0] > 1e-4) & (lean.resid[:3, 0] < 8e-3))


def test_two_tail_instances_per_wave_give_the_same_verdicts(emu, track):
    """The default form of the tail solver - two instances per wave, 32 lanes each, three entries per lane, phase 1 through
    ipm3<SOFT> - against its one-instance-per-wave form (mpmpc_set_tail_kernel(h, 2) / emu_set_lean_tail(2): the split layout,
    the general routine): the same statuses and iteration counts; the least-violation points, rays and relaxed plans agree to
    ~1e-7 (phase 1 converges to 1e-11 in the scaled problem along a different arithmetic path), certified optima to 1e-9."""
    for cfgid, B, accept in ((4, 513, 1), (5, 300, 1), (4, 257, 0)):
        sc = scenarios.make(cfgid, track, B=B)
        cfg = T.stock_config(sc.N, sc.weights)
        qp = emu.assemble(cfg, track, _inputs(sc))
        st = mpmpc.default_settings(phase1_accept=accept)
        two, n_tail2 = emu.solve_launch(cfg, st, qp, G=32)
        assert emu.lib.emu_last_tail2() == 0
        try:
            emu.lib.emu_set_lean_tail(2)
            one, n_tail = emu.solve_launch(cfg, st, qp, G=32)
            assert emu.lib.emu_last_tail2() == 0
        finally:
            emu.lib.emu_set_lean_tail(1)
        assert n_tail == n_tail2 and n_tail >= 15
        assert np.array_equal(one.status, two.status) and np.array_equal(one.iters[:, 0], two.iters[:, 0])
        assert np.max(np.abs(one.iters[:, 1] - two.iters[:, 1])) <= 1
        ok = one.status == 1
        np.testing.assert_allclose(two.z[ok], one.z[ok], rtol=0, atol=1e-9)
        np.testing.assert_allclose(two.z, one.z, rtol=0, atol=1e-6)
        np.testing.assert_allclose(two.resid[:, 0], one.resid[:, 0], rtol=1e-6, atol=1e-9)
        inf = np.flatnonzero(two.status == mpmpc.PRIMAL_INFEASIBLE)
        for i in inf[:12]:
            assert _farkas_ok(qp[:, i, :], sc.N, two.y[i])


@pytest.mark.parametrize("N,B", [(40, 160), (50, 96), (32, 120)])
def test_reduced_native_tail_solver_at_horizons_above_31(N, B, emu, track):
    """Horizons 32 .. 63: one lane per stage, one instance per wave (<64, 32>), phase 1 through ipm3<SOFT>.  Against the general
    kernel on the whole tail: the same statuses and ADMM counters, certified optima to 1e-9, least-violation points and relaxed
    plans to 1e-6; nothing left to the general kernel."""
    sc = scenarios.make(4, track, B=B, N=N)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, _inputs(sc))
    st = mpmpc.default_settings()
    try:
        emu.lib.emu_set_lean_tail(0)
        gen, n_tail = emu.solve_launch(cfg, st, qp, G=64)
    finally:
        emu.lib.emu_set_lean_tail(1)
    lean, n_tail2 = emu.solve_launch(cfg, st, qp, G=64)
    assert n_tail == n_tail2 and n_tail >= 3 and emu.lib.emu_last_tail2() == 0
    assert np.array_equal(lean.status, gen.status) and np.array_equal(lean.iters[:, 0], gen.iters[:, 0])
    assert (lean.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 2
    ok = lean.status == 1
    np.testing.assert_allclose(lean.z[ok], gen.z[ok], rtol=0, atol=1e-9)
    np.testing.assert_allclose(lean.z, gen.z, rtol=0, atol=1e-6)
    for i in np.flatnonzero(lean.status == mpmpc.PRIMAL_INFEASIBLE)[:8]:
        assert _farkas_ok(qp[:, i, :], sc.N, lean.y[i])
