"""Horizons above 63 (the reference has no limit, src/MPC.py:73-74).  Up to N = 63 an instance is a part
of / one 64-lane wavefront; beyond that it takes a WORKGROUP of 2 / 4 wavefronts - still one lane per stage, the lanes of
different wavefronts talk through LDS (csrc/lane_gpu.hpp: LaneBlock, mpmpc_solve_block_kernel).  Same lane code: the CPU suite
runs it on an emulated group of 128 / 256 lanes (tests/emul/emul_wide.cpp) against the C oracle and the plain-numpy KKT test;
the GPU suite runs libmpmpc.so against that emulation, the C oracle and the certificates."""
import numpy as np
import pytest

import mpmpc
import mpmpc_testlib as T
import oracle_c as OC
import osqp_np as O
import scenarios

LONG = [(64, 2), (64, 4), (100, 4), (127, 2), (128, 4), (200, 2), (255, 4)]


def _oracle(track, sc, weights, xmin=scenarios.XMIN, xmax=scenarios.XMAX, **st):
    ocfg = OC.mpc_cfg(sc.N, weights, scenarios.UMIN, scenarios.UMAX, xmin, xmax, 4.0, 0.12)
    return OC.mpc_batch(ocfg, OC.settings(**st), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)


@pytest.mark.parametrize("N,cfgid", LONG)
def test_long_horizons_in_the_emulation_against_the_c_oracle(N, cfgid, emu, track):
    """K1 + the general solver on an emulated workgroup: statuses and ADMM counts of the C oracle, controls to 1e-6 (measured
    1e-13), every solved instance through the vectorised plain-numpy KKT test on K1's own output, Farkas rays for the rest."""
    B = 10
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(cfgid, tw, B=B, N=N)
    cfg = T.stock_config(N, sc.weights)
    assert mpmpc.stage_ld(N) == (128 if N + 1 <= 128 else 256) if N + 1 > 64 else True
    qp = emu.assemble(cfg, tw, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    st = mpmpc.default_settings(phase1_accept=0)
    sol = emu.solve(cfg, st, qp)
    ref = _oracle(track, sc, scenarios.WEIGHTS[sc.weights])
    assert np.array_equal(sol.status, ref["status"]) and np.array_equal(sol.iters[:, 0], ref["iters"][:, 0])
    ok = sol.status == 1
    assert ok.sum() >= B // 2 and np.max(np.abs(sol.u0[ok] - ref["u0"][ok])) <= 1e-6
    e = np.abs(sol.z[ok] - ref["z"][ok])
    e[:, -1] = 0.0                                   # the cost-free kappa_{N-1}
    e[:, 3 * N + 1] = 0.0                            # ... and e_psi,N it drives
    assert e.max() <= 1e-6
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, sol.z[ok], sol.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    if (~ok).any():
        good, _, _ = T.farkas_batch(qp[:, ~ok, :], N, sol.y[~ok])
        assert good.all()


@pytest.mark.parametrize("N", [65, 80, 96, 112, 126, 129, 144, 160, 192, 224, 254])
def test_cyclic_reduction_of_partly_filled_rows(N, emu, track):
    """The chains of a workgroup instance are right-aligned: their first rows are empty or partly filled, differently for every
    horizon.  The reduced-native workgroup solver (cyclic reduction inside the rows, the row survivors in turn, one step across
    the wavefronts at 256 lanes) against the C oracle at horizons on and beside the row boundaries."""
    B = 6
    tw = T.wide_track(track, emu, N)
    for cfgid in (2, 4):
        sc = scenarios.make(cfgid, tw, B=B, N=N)
        cfg = T.stock_config(N, sc.weights)
        qp = emu.assemble(cfg, tw, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
        sol = emu.solve(cfg, mpmpc.default_settings(phase1_accept=0), qp)
        ref = _oracle(track, sc, scenarios.WEIGHTS[sc.weights])
        assert np.array_equal(sol.status, ref["status"])
        ok = sol.status == 1
        assert ok.any() and np.max(np.abs(sol.u0[ok] - ref["u0"][ok])) <= 1e-6
        prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, sol.z[ok], sol.y[ok])
        assert max(prim.max(), stat.max(), comp.max()) <= 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("N", [70, 150])
def test_long_horizon_general_variants_on_device(N, track, emu):
    """The workgroup kernel's OTHER variants on the device - full weight matrices (3 x 3 dense blocks), the time-optimal
    weights and bounded e_psi / t (the full 3-state problem), and the restated OSQP alone - at 128 and at 256 lanes (where
    the sweeps of the factorisation are staged wavefront by wavefront).  EVERY instance of a batch of 96 against the C ORACLE
    (statuses, controls to 1e-6, the plan to 1e-6) and through the plain-numpy KKT / Farkas tests; eight of
    them against the emulation of the same lane code (statuses, iteration counts, z to 1e-9)."""
    from test_emul_parity import full_weight_config
    B = 96
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(4, tw, B=B, N=N)
    inp = (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    xmin, xmax = np.array([-np.inf, -0.6, -np.inf]), np.array([np.inf, 0.6, 0.2 * N])
    Qt, Rt, QNt = scenarios.WEIGHTS["time_optimal"]
    Qs, Rs, QNs = scenarios.WEIGHTS["stock"]
    strict = mpmpc.default_settings(phase1_accept=0)          # (every proven infeasibility reported: the oracle's semantics)
    cases = [("full weights", full_weight_config(N, "full", max_batch=B), strict, None),
             ("time-optimal", mpmpc.make_config(N, Qt, Rt, QNt, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, 4.0, 0.12, max_batch=B),
              strict, ((Qt, Rt, QNt), scenarios.XMIN, scenarios.XMAX)),
             ("bounded states", mpmpc.make_config(N, Qs, Rs, QNs, xmin, xmax, scenarios.UMIN, scenarios.UMAX, 4.0, 0.12, max_batch=B),
              strict, ((Qs, Rs, QNs), xmin, xmax)),
             ("stock OSQP", T.stock_config(N, max_batch=B), mpmpc.stock_settings(), None)]
    for name, cfg, st, orc in cases:
        h = mpmpc.Handle(cfg, st)
        h.set_path(track.kappa, track.v_ref, track.ds_next)
        qp = h.assemble(*inp)
        sol = h.solve(*inp, want_y=True)
        h.close()
        nn = 8
        ref = emu.solve(cfg, st, np.ascontiguousarray(qp[:, :nn, :]))
        assert np.array_equal(sol.status[:nn], ref.status), (name, sol.status[:nn], ref.status)
        assert np.array_equal(sol.iters[:nn, 0], ref.iters[:, 0]), name
        assert np.max(np.abs(sol.iters[:nn, 1] - ref.iters[:, 1])) <= 1, name
        okr = ref.status == 1
        assert okr.any() and np.max(np.abs(sol.z[:nn][okr] - ref.z[okr])) <= (1e-9 if name != "stock OSQP" else 1e-7), name
        if name == "stock OSQP":
            r = _oracle(track, sc, scenarios.WEIGHTS["stock"], polish=0, early_polish=0, phase1=0)
            assert np.array_equal(sol.status, r["status"]) and np.array_equal(sol.iters[:, 0], r["iters"][:, 0]), name
            ok = sol.status == 1
            assert np.max(np.abs(sol.z[ok] - r["z"][ok])) <= 1e-6
            continue
        ok = sol.status == 1
        assert ok.sum() >= B // 2, name
        if orc is not None:
            # diagonal weights: the C oracle's MPC path on the same inputs, all 96 instances
            r = _oracle(track, sc, *orc)
            assert np.array_equal(sol.status, r["status"]), (name, np.flatnonzero(sol.status != r["status"]))
            assert np.max(np.abs(sol.u0[ok] - r["u0"][ok])) <= 1e-6, name
            e = np.abs(sol.z[ok] - r["z"][ok])
            e[:, -1] = 0.0
            e[:, 3 * N + 1] = 0.0                        # (kappa_{N-1} and the e_psi,N it drives carry no cost)
            assert e.max() <= 1e-6, name
            prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, sol.z[ok], sol.y[ok])
            assert max(prim.max(), stat.max(), comp.max()) <= 1e-8, name
            if (~ok).any():
                good, _, _ = T.farkas_batch(qp[:, ~ok, :], N, sol.y[~ok])
                assert good.all(), name
        else:
            # full weights: the C oracle's generic QP path on the dense (P, q, A, l, u) rebuilt from K1's fields, instance by instance
            for i in range(B):
                P, q, A, l, u = T.qp_to_dense_full(qp[:, i, :], N, cfg)
                x, y, info = OC.solve(P, q, A, l, u)
                assert sol.status[i] == info.status, (name, i, sol.status[i], info.status)
                if sol.status[i] == 1:
                    assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8), (name, i)
                    if info.polished == 1:
                        e = np.abs(sol.z[i] - x)
                        e[-1] = 0.0
                        assert e.max() <= 1e-6 and np.max(np.abs(sol.u0[i] - [x[3 * (N + 1)], np.arctan(x[3 * (N + 1) + 1] * scenarios.CAR_LENGTH)])) <= 1e-6, (name, i)
                else:
                    assert O.farkas_certificate(A, l, u, sol.y[i], 1e-6)["ok"], (name, i)


@pytest.mark.parametrize("N", [64, 130])
def test_long_horizons_other_problem_classes_in_the_emulation(N, emu, track):
    """What else the general kernel serves, at a horizon that needs a workgroup: full weight matrices (dense stage blocks),
    the time-optimal weights, bounded e_psi / t, and the restated OSQP alone (stock settings) - against the dense / C oracle."""
    from test_emul_parity import full_weight_config
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(4, tw, B=4, N=N)
    inp = (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    # full weights
    cfg = full_weight_config(N, "full")
    qp = emu.assemble(cfg, tw, inp)
    sol = emu.solve(cfg, mpmpc.default_settings(), qp)
    for i in range(sc.B):
        P, q, A, l, u = T.qp_to_dense_full(qp[:, i, :], N, cfg)
        x, y, info = OC.solve(P, q, A, l, u)
        assert sol.status[i] == info.status or (sol.status[i] == 2 and info.status == -3)
        if info.status == 1 and info.polished == 1:
            e = np.abs(sol.z[i] - x)
            e[-1] = 0.0
            assert e.max() <= 1e-6 and O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
    # time-optimal weights, and a box on e_psi and t
    xmin, xmax = np.array([-np.inf, -0.6, -np.inf]), np.array([np.inf, 0.6, 0.2 * N])
    for weights, lo, hi in (("time_optimal", scenarios.XMIN, scenarios.XMAX), ("stock", xmin, xmax)):
        Q, R, QN = scenarios.WEIGHTS[weights]
        cfg = mpmpc.make_config(N, Q, R, QN, lo, hi, scenarios.UMIN, scenarios.UMAX, 4.0, 0.12)
        sol = emu.solve(cfg, mpmpc.default_settings(phase1_accept=0), emu.assemble(cfg, tw, inp))
        ref = _oracle(track, sc, (Q, R, QN), lo, hi)
        assert np.array_equal(sol.status, ref["status"])
        ok = sol.status == 1
        assert ok.any() and np.max(np.abs(sol.u0[ok] - ref["u0"][ok])) <= 1e-6
    # the restated OSQP alone: its verdicts and iteration counts
    cfg = T.stock_config(N)
    sol = emu.solve(cfg, mpmpc.stock_settings(), emu.assemble(cfg, tw, inp))
    ref = _oracle(track, sc, scenarios.WEIGHTS["stock"], polish=0, early_polish=0, phase1=0)
    assert np.array_equal(sol.status, ref["status"]) and np.array_equal(sol.iters[:, 0], ref["iters"][:, 0])
    ok = sol.status == 1
    assert np.max(np.abs(sol.z[ok] - ref["z"][ok])) <= 1e-6 if ok.any() else True


@pytest.mark.parametrize("N,cfgid,B", [(64, 2, 10), (64, 4, 12), (65, 4, 6), (80, 4, 16), (96, 2, 6), (100, 4, 10), (112, 2, 6), (126, 4, 6), (127, 2, 10)])
def test_two_stages_per_lane_in_one_wavefront_against_the_workgroup_emulation(N, cfgid, B, emu, track):
    """Horizons 64 .. 127 (round 6): the reduced-native solver with TWO stages per lane - the whole instance in one wavefront, a
    chain of four rows (csrc/lane_pair.hpp, mpmpc_solver_s2.hpp: level H inside the lanes, four in-row levels, the row survivors
    in turn) - against the workgroup emulation of the same solver (one stage per lane, 128 lanes) and the C oracle: statuses,
    iteration counts, controls; KKT with plain numpy.  Row boundaries (N = 64, 65, 96, 112, 126, 127) included."""
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(cfgid, tw, B=B, N=N)
    cfg = T.stock_config(N, sc.weights)
    st = mpmpc.default_settings(phase1_accept=0)
    qp = emu.assemble(cfg, tw, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    pair = emu.solve(cfg, st, qp)                  # the launcher's sequence: pair kernel, then the workgroup kernel on its list
    wg = emu.solve(cfg, st, qp, G=128)             # ... and with mpmpc_set_packing(h, 128): the workgroup kernels alone
    assert np.array_equal(pair.status, wg.status) and np.array_equal(pair.iters, wg.iters)
    ok = pair.status == 1
    assert ok.sum() >= B // 2 and np.max(np.abs(pair.u0[ok] - wg.u0[ok])) <= 1e-13 and np.max(np.abs(pair.z[ok] - wg.z[ok])) <= 1e-10
    ref = _oracle(track, sc, scenarios.WEIGHTS[sc.weights])
    assert np.array_equal(pair.status, ref["status"]) and np.max(np.abs(pair.u0[ok] - ref["u0"][ok])) <= 1e-6
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, pair.z[ok], pair.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    if (~ok).any():
        good, _, _ = T.farkas_batch(qp[:, ~ok, :], N, pair.y[~ok])
        assert good.all()
    # ... and with the DEFAULT verdict semantics (marginally infeasible instances come back as plans over relaxed boxes, status 2):
    # the tail solver on the pair layout (mpmpc_reduced_tail_pair_kernel<64>) against the general workgroup kernel's phase 1
    dflt = mpmpc.default_settings()
    pair2, wg2 = emu.solve(cfg, dflt, qp), emu.solve(cfg, dflt, qp, G=128)
    assert np.array_equal(pair2.status, wg2.status)
    u = pair2.status > 0
    assert np.max(np.abs(pair2.u0[u] - wg2.u0[u])) <= 1e-6 and np.max(np.abs(pair2.resid[:, 0] - wg2.resid[:, 0])) <= 1e-7


@pytest.mark.parametrize("N,cfgid,B", [(128, 2, 6), (128, 4, 8), (129, 2, 4), (150, 4, 16), (191, 4, 6), (192, 2, 4), (200, 2, 6), (254, 2, 4), (255, 4, 8)])
def test_two_stages_per_lane_on_a_workgroup_of_128_lanes(N, cfgid, B, emu, track):
    """Horizons 128 .. 255 (round 6): the pair layout on a workgroup of TWO wavefronts - one chain of eight rows, the step from
    row 3 to row 4 crossing the wavefronts through LDS - with the LEAN cold storage of the device kernel (37 pair slots:
    ReducedSolver::kLean), against the 256-lane workgroup emulation of the same solver and the C oracle.  The loose first
    tolerance of the second run forces repeated attempts: the parked iterate comes back from the slots it shares."""
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(cfgid, tw, B=B, N=N)
    cfg = T.stock_config(N, sc.weights)
    qp = emu.assemble(cfg, tw, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    ref = _oracle(track, sc, scenarios.WEIGHTS[sc.weights])
    for kw in (dict(), dict(native_ipm_tol=3e-5), dict(native_ipm_tol=1e-2)):
        st = mpmpc.default_settings(phase1_accept=0, **kw)
        pair = emu.solve(cfg, st, qp)                  # the launcher's sequence: K2rb2, then the 256-lane workgroup kernel on its list
        wg = emu.solve(cfg, st, qp, G=256)             # mpmpc_set_packing(h, 256): round 5's kernels alone
        assert np.array_equal(pair.status, wg.status) and np.array_equal(pair.iters, wg.iters)
        ok = pair.status == 1
        assert ok.sum() >= B // 2 and np.max(np.abs(pair.u0[ok] - wg.u0[ok])) <= 1e-13 and np.max(np.abs(pair.z[ok] - wg.z[ok])) <= 1e-10
        assert np.array_equal(pair.status, ref["status"]) and np.max(np.abs(pair.u0[ok] - ref["u0"][ok])) <= 1e-6
        prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, pair.z[ok], pair.y[ok])
        assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
        if (~ok).any():
            good, _, _ = T.farkas_batch(qp[:, ~ok, :], N, pair.y[~ok])
            assert good.all()
    # the default verdict semantics: the tail solver on the same workgroup layout (mpmpc_reduced_tail_pair_block_kernel) against
    # the general kernel's phase 1
    dflt = mpmpc.default_settings()
    pair2, wg2 = emu.solve(cfg, dflt, qp), emu.solve(cfg, dflt, qp, G=256)
    assert np.array_equal(pair2.status, wg2.status) and np.max(np.abs(pair2.resid[:, 0] - wg2.resid[:, 0])) <= 1e-7
    # (at 1e-2 the first attempt of an obstacle-course instance stops too early for its active-set rounds and is taken up again at
    #  1e-4 from the iterate the lean storage parked - measured: 6 - 7 iterations where 3e-5 takes 5; the lean and the full storage
    #  give the same iterates bit for bit, which is what the equality above checks)


@pytest.mark.parametrize("N,B", [(64, 8), (70, 12), (96, 6), (100, 8), (127, 8), (128, 4), (150, 6), (255, 4)])
def test_terminal_time_kernel_with_two_stages_per_lane_at_long_horizons(N, B, emu, track):
    """Time-optimal weights (a terminal cost on the time state, README.md:56 of the reference) at horizons 64 .. 255: the
    terminal-time reduced-native solver (2 x 2 blocks + Sherman-Morrison) on the pair layout - one wavefront per instance up to
    127, a workgroup of two above - instead of the general 3-state solver on a workgroup: against that solver's emulation and
    the C oracle (statuses, controls to 1e-6, measured 2e-10), KKT with plain numpy on the FULL problem."""
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(3, tw, B=B, N=N)
    cfg = T.stock_config(N, "time_optimal")
    st = mpmpc.default_settings(phase1_accept=0)
    qp = emu.assemble(cfg, tw, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    pair = emu.solve(cfg, st, qp)                  # the launcher's sequence
    wg = emu.solve(cfg, st, qp, G=128 if N < 128 else 256)      # the general workgroup kernel (mpmpc_set_packing(h, 128 / 256))
    assert np.array_equal(pair.status, wg.status)
    ok = pair.status == 1
    assert ok.sum() >= B // 2 and np.max(np.abs(pair.u0[ok] - wg.u0[ok])) <= 1e-8
    # (fewer iterations than the general kernel: its own Ruiz pass and start, as at N = 50)
    assert pair.iters[ok, 1].mean() < wg.iters[ok, 1].mean()
    ref = _oracle(track, sc, scenarios.WEIGHTS["time_optimal"])
    assert np.array_equal(pair.status, ref["status"]) and np.max(np.abs(pair.u0[ok] - ref["u0"][ok])) <= 1e-6
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, pair.z[ok], pair.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8


def test_horizon_limits_are_checked():
    with pytest.raises(ValueError):
        T.stock_config(256)
    with pytest.raises(ValueError):
        T.stock_config(2)
    assert mpmpc.MAX_HORIZON == 255 and [mpmpc.stage_ld(n) for n in (15, 16, 31, 63, 64, 127, 128, 255)] == [16, 32, 32, 64, 128, 128, 256, 256]


# ------------------------------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("N,cfgid", LONG)
def test_long_horizons_on_device(N, cfgid, track, emu):
    """libmpmpc.so at horizons 64 .. 255: K1 bit for bit with the emulation, the workgroup kernel with its emulation (statuses,
    iteration counts, z to 1e-9), the whole batch against the C oracle (statuses, controls to 1e-6) and through the KKT /
    Farkas tests with plain numpy."""
    B = 96
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(cfgid, tw, B=B, N=N)
    cfg = T.stock_config(N, sc.weights, max_batch=B)
    st = mpmpc.default_settings(phase1_accept=0)
    h = mpmpc.Handle(cfg, st)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    inp = (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    qp = h.assemble(*inp)
    sol = h.solve(*inp, want_y=True)
    # ... the resident path and the corridor table instead of rows give the same bits
    h.set_corridor(tw.ub_obstacles if sc.obstacles else tw.ub_free, tw.lb_obstacles if sc.obstacles else tw.lb_free)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev)
    h.solve_resident(B)
    sol2 = h.download(B, want_y=True)
    h.close()
    assert np.array_equal(sol.z, sol2.z) and np.array_equal(sol.status, sol2.status) and np.array_equal(sol.y, sol2.y)
    qp_e = emu.assemble(cfg, tw, inp)
    cap = 15                                   # the speed cap goes through tan(): device libm vs the host's, a few ulp
    other = np.delete(np.arange(mpmpc.NUM_FIELDS), cap)
    assert np.array_equal(qp[other][:, :, :N + 1], qp_e[other][:, :, :N + 1])
    assert np.max(np.abs(qp[cap, :, :N + 1] - qp_e[cap, :, :N + 1])) <= 8 * np.finfo(float).eps
    nn = 12
    ref = emu.solve(cfg, st, np.ascontiguousarray(qp[:, :nn, :]))
    assert np.array_equal(sol.status[:nn], ref.status) and np.array_equal(sol.iters[:nn], ref.iters)
    okr = ref.status == 1
    assert np.max(np.abs(sol.z[:nn][okr] - ref.z[okr])) <= 1e-9
    orc = _oracle(track, sc, scenarios.WEIGHTS[sc.weights])
    assert np.array_equal(sol.status, orc["status"])
    ok = sol.status == 1
    assert ok.mean() > 0.5 and np.max(np.abs(sol.u0[ok] - orc["u0"][ok])) <= 1e-6
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], N, sol.z[ok], sol.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    if (~ok).any():
        good, _, _ = T.farkas_batch(qp[:, ~ok, :], N, sol.y[~ok])
        assert good.all()


@pytest.mark.gpu
@pytest.mark.parametrize("N,cfgid", [(64, 4), (100, 2), (127, 4), (128, 4), (200, 2), (255, 4)])
def test_pair_kernel_and_workgroup_kernel_agree_on_device(N, cfgid, track, emu):
    """Horizons 64 .. 255: the default (two stages per lane: one wavefront per instance up to 127, a workgroup of two above)
    against mpmpc_set_packing(h, 128 / 256) (the one-stage workgroup kernels of round 5) on 512 instances: statuses, iteration
    counts of the certified instances, controls to 1e-12."""
    B = 512
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(cfgid, tw, B=B, N=N)
    cfg = T.stock_config(N, sc.weights, max_batch=B)
    sols = {}
    for lanes in (0, 128 if N < 128 else 256):
        h = mpmpc.Handle(cfg, mpmpc.default_settings())
        h.set_path(track.kappa, track.v_ref, track.ds_next)
        h.set_packing(lanes)
        sols[lanes] = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
        h.close()
    a, b = sols[0], sols[128 if N < 128 else 256]
    assert np.array_equal(a.status, b.status)
    ok = a.status == 1
    assert ok.mean() > 0.5 and np.max(np.abs(a.iters[ok, 1] - b.iters[ok, 1])) <= 1 and (a.iters[ok, 1] != b.iters[ok, 1]).sum() <= 2
    assert np.max(np.abs(a.u0[ok] - b.u0[ok])) <= 1e-12


@pytest.mark.gpu
def test_long_horizon_full_weights_and_host_class_on_device(track, emu):
    """Full weight matrices at N = 100 through the workgroup kernel (device = emulation, KKT on the dense data), and the
    reference's own loop (MPC.get_control) at N = 80 against the emulation-backed controller, step by step."""
    from test_emul_parity import full_weight_config
    import test_host_mpc as H
    from spatial_bicycle_models import TemporalState
    N, B = 100, 48
    tw = T.wide_track(track, emu, N)
    sc = scenarios.make(4, tw, B=B, N=N)
    cfg = full_weight_config(N, "full", max_batch=B)
    h = mpmpc.Handle(cfg)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    inp = (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    qp = h.assemble(*inp)
    sol = h.solve(*inp, want_y=True)
    h.close()
    ref = emu.solve(cfg, mpmpc.default_settings(), np.ascontiguousarray(qp[:, :8, :]))
    assert np.array_equal(sol.status[:8], ref.status) and np.max(np.abs(sol.z[:8][ref.status == 1] - ref.z[ref.status == 1])) <= 1e-9
    for i in np.flatnonzero(sol.status == 1)[:10]:
        P, q, A, l, u = T.qp_to_dense_full(qp[:, i, :], N, cfg)
        assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
    # the host class: ten steps of the reference's loop at N = 80, the device against the emulation-backed controller
    g = np.load(H.G + "/g6_closed_loop_N30.npz")
    us = []
    for backend in (None, "emu"):
        m, rp, car = H.build_world()
        from scipy import sparse
        Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
        ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
        scn = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
        from MPC import MPC
        be = T.EmuBackend(T.stock_config(80), mpmpc.default_settings()) if backend == "emu" else None
        mpc = MPC(car, 80, Q, R, QN, scn, ic, 4.0, backend=be)
        out = []
        for t in range(0, 40, 4):
            car.s = float(g["s"][t])
            car.temporal_state = TemporalState(*g["pose"][t])
            mpc.current_control = np.zeros(160)
            mpc.infeasibility_counter = 0
            out.append((mpc.get_control(), mpc.last_status))
        us.append(out)
    for (u_d, s_d), (u_e, s_e) in zip(*us):
        assert s_d == s_e and np.max(np.abs(u_d - u_e)) <= 1e-9
