"""-m gpu: the HIP path through the C ABI (libmpmpc.so) against the oracle, on a real MI355X."""
import numpy as np
import pytest

import mpc_np as M
import mpmpc
import mpmpc_testlib as T
import osqp_np as O
import scenarios

pytestmark = pytest.mark.gpu


# verdict semantics of the oracle and of the certified-mode goldens (G5, G6): every proven infeasibility is reported.  The
# library's default hands marginal ones back as usable plans (mpmpc_settings::phase1_accept), pinned by golden G6s.
STRICT = dict(phase1_accept=0)


def _handle(track, N, weights, max_batch, settings=None, table=None):
    cfg = T.stock_config(N, weights, max_batch=max_batch)
    h = mpmpc.Handle(cfg, settings or mpmpc.default_settings())
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    if table == "free":
        h.set_corridor(track.ub_free, track.lb_free)
    elif table == "obstacles":
        h.set_corridor(track.ub_obstacles, track.lb_obstacles)
    return h


def test_device_present():
    assert mpmpc.device_count() >= 1


def test_library_on_the_gpu_box_is_the_trees():
    """The .so that travelled here was built from exactly these sources (hash compiled into mpmpc_version())."""
    import __graft_entry__ as g
    assert g.source_hash().encode() in mpmpc.load_library().mpmpc_version()


@pytest.mark.parametrize("cfgid,B,N", [(2, 40, 30), (4, 40, 30), (3, 12, 50), (2, 12, 10), (4, 12, 3)])
def test_k1_assembly_bit_exact(cfgid, B, N, track, otrack):
    sc = scenarios.make(cfgid, track, B=B, N=N)
    h = _handle(track, N, sc.weights, B, table="obstacles" if sc.obstacles else "free")
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    w = M.Weights.time_optimal() if sc.weights == "time_optimal" else M.Weights.stock()
    for i in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], N)
        P0, q0, A0, l0, u0 = M.assemble(otrack, int(sc.wp_id[i]), sc.x0[i], sc.cc_prev[i], sc.lb[i], sc.ub[i], N,
                                        w, M.Limits.stock())
        assert np.array_equal(A, A0) and np.array_equal(Pd, np.diag(P0)) and np.array_equal(q, q0)
        assert np.array_equal(l, l0)
        fin = np.isfinite(u0)
        assert np.array_equal(u[~fin], u0[~fin])
        # speed cap goes through tan()/sqrt(): device libm vs numpy, a few ulp
        assert np.max(np.abs(u[fin] - u0[fin])) <= 8 * np.finfo(float).eps
    qp_t = h.assemble(sc.wp_id, sc.x0, sc.cc_prev)          # corridor from the resident table
    assert np.array_equal(qp, qp_t)
    h.close()


def test_k1_matches_reference_capture(track):
    g = np.load(M.GOLDEN + "/g4_assembly_N30.npz")
    N, B = 30, g["s"].size
    h = _handle(track, N, "stock", B)
    qp = h.assemble(g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"])
    for c in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, c, :], N)
        assert np.array_equal(q, g["q"][c]) and np.array_equal(l, g["l"][c]) and np.array_equal(Pd, g["P_diag"][c])
        fin = np.isfinite(g["u"][c])
        assert np.max(np.abs(u[fin] - g["u"][c][fin])) <= 8 * np.finfo(float).eps
    h.close()


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_device_reaches_the_g5_optima_of_the_reference_qps(N, track):
    """The reference's own captured inputs (G4) through the C ABI: statuses of G5 and its certified optima to 1e-6
    (the north-star tolerance), first control to 1e-8."""
    g4 = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    g5 = np.load(M.GOLDEN + "/g5_solutions_N%d.npz" % N)
    B = g4["s"].size
    h = _handle(track, N, str(g4["weights"][0]), B)
    sol = h.solve(g4["wp_id"].astype(np.int32), g4["x0"], g4["cc_prev"], g4["lb"], g4["ub"], want_y=True)
    h.close()
    assert np.array_equal(sol.status, g5["status"])
    ok = g5["status"] == 1
    assert np.max(np.abs(sol.z[ok] - g5["x"][ok])) < 1e-6
    assert np.max(np.abs(sol.z[ok][:, -2 * N:-2 * N + 2] - g5["x"][ok][:, -2 * N:-2 * N + 2])) < 1e-8
    L = 0.12
    assert np.max(np.abs(sol.u0[ok, 1] - np.arctan(g5["x"][ok][:, -2 * N + 1] * L))) < 1e-8


def test_admm_stock_matches_oracle(track):
    """stock OSQP settings, no polish: the reference's own solver call (src/MPC.py:159,183)"""
    sc = scenarios.make(2, track, B=24)
    h = _handle(track, sc.N, sc.weights, sc.B, mpmpc.default_settings(polish=0))
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    for i in range(sc.B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
        r = O.solve(np.diag(Pd), q, A, l, u, O.Settings())
        assert sol.status[i] == r.status and sol.iters[i, 0] == r.iters
        assert np.max(np.abs(sol.z[i] - r.x)) < 1e-7 and np.max(np.abs(sol.y[i] - r.y)) < 1e-7
    h.close()


@pytest.mark.parametrize("cfgid,B,N", [(2, 48, 30), (4, 48, 30), (3, 12, 50), (2, 16, 10), (4, 16, 3), (4, 12, 31), (2, 8, 32)])
def test_certified_matches_oracle(cfgid, B, N, track):
    """max |u - u_ref| <= 1e-6 with u_ref the oracle's KKT-certified optimum"""
    sc = scenarios.make(cfgid, track, B=B, N=N)
    h = _handle(track, N, sc.weights, B)
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    L = scenarios.CAR_LENGTH
    n_cert = n_inf = n_skip = 0
    for i in range(B):
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], N)
        r = O.solve(np.diag(Pd), q, A, l, u, O.Settings(polish=2))
        if r.polished != 1 and sol.status[i] == 1:
            assert O.kkt_certificate(np.diag(Pd), q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
            n_skip += 1
            continue
        # (default verdicts on the device: an instance the oracle proves infeasible may be MARGINALLY so - status 2, a plan on
        #  relaxed boxes, mpmpc_settings::phase1_accept)
        if sol.status[i] == mpmpc.SOLVED_INACCURATE and r.status == O.PRIMAL_INFEASIBLE:
            assert sol.resid[i, 0] <= 1e-3 + 1e-3 * scenarios.UMAX[1] * 1.01
            n_inf += 1
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
            e[-1] = 0.0
            e[3 * N + 1] = 0.0
            assert e.max() <= 1e-6
            assert O.kkt_certificate(np.diag(Pd), q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
    # every instance is accounted for: compared with the oracle's certified optimum, or infeasible on both sides; on these
    # batches the dense numpy oracle never gives up ("weak" 4: the test used to ask for half)
    assert n_skip == 0 and n_cert + n_inf == B, (n_cert, n_inf, n_skip)
    h.close()


@pytest.mark.parametrize("cfgid,B", [(2, 1024), (4, 2048)])
def test_full_size_against_emulation_and_certificates(cfgid, B, track, emu):
    """BASELINE sizes: GPU == lock-step CPU emulation of the same source; every solved instance
    carries an independent KKT certificate; infeasible ones are flagged, never silently solved."""
    sc = scenarios.make(cfgid, track, B=B)
    h = _handle(track, sc.N, sc.weights, B)
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    # (solve_launch: what the launcher runs - the reduced-native kernel, then the general kernel on its tail)
    ref, _ = emu.solve_launch(T.stock_config(sc.N, sc.weights), h.settings, qp, G=64)
    assert np.array_equal(sol.status, ref.status)
    assert np.array_equal(sol.iters[:, 0], ref.iters[:, 0])
    ok = sol.status == 1
    assert np.max(np.abs(sol.u0[ok] - ref.u0[ok])) <= 1e-9
    worst = 0.0
    for i in np.flatnonzero(ok)[:256]:
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
        c = O.kkt_certificate(np.diag(Pd), q, A, l, u, sol.z[i], sol.y[i])
        worst = max(worst, c["prim"], c["stat"], c["comp"])
    assert worst <= 1e-8
    assert ok.mean() > 0.8
    if cfgid == 4:
        assert (sol.status == mpmpc.PRIMAL_INFEASIBLE).sum() > 0
    h.close()


def test_ragged_and_single_instance(track):
    sc = scenarios.make(2, track, B=7, N=10)
    h = _handle(track, 10, "stock", 16)
    full = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    one = h.solve(sc.wp_id[3:4], sc.x0[3:4], sc.cc_prev[3:4], sc.lb[3:4], sc.ub[3:4])
    assert one.status[0] == full.status[3] and np.array_equal(one.z[0], full.z[3])
    h.close()


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_horizons_weights_and_batches_against_emulation(seed, track, emu):
    """Random (horizon, weighting / corridor kind, batch size, shuffled and perturbed poses): the device and the
    lock-step emulation of the same lane code agree in status, iteration counts and, to 1e-9, in z and u0 - for
    every lane packing the horizon allows (profiles/stress.py is the long form of this sweep)."""
    rng = np.random.default_rng(seed)
    for _ in range(10):
        N = int(rng.choice([3, 4, 7, 10, 15, 16, 17, 24, 30, 31, 32, 33, 40, 45, 50]))
        cfg_id = int(rng.choice([2, 3, 4]))
        B = int(rng.integers(1, 24))
        weights = scenarios.CONFIGS[cfg_id]["weights"]
        sc = scenarios.make(cfg_id, track, B=B, N=N)
        perm = rng.permutation(B)
        wp, x0, cc, lb, ub = sc.wp_id[perm], sc.x0[perm], sc.cc_prev[perm], sc.lb[perm], sc.ub[perm]
        x0 = x0 + rng.normal(0, 0.01, x0.shape) * np.array([1.0, 1.0, 0.0])
        h = _handle(track, N, weights, B)
        dev = h.solve(wp, x0, cc, lb, ub, want_y=True)
        h.close()
        cfg = T.stock_config(N, weights, max_batch=B)
        qp = emu.assemble(cfg, track, (wp, x0, cc, lb, ub))
        for G in sorted({64, 32 if N + 1 <= 32 else 64, 16 if N + 1 <= 16 else 64}):
            ref, _ = emu.solve_launch(cfg, mpmpc.default_settings(), qp, G=G)      # the launcher's own sequence of kernels
            assert np.array_equal(dev.status, ref.status) and np.array_equal(dev.iters, ref.iters), (N, cfg_id, B, G)
            ok = dev.status == 1
            if ok.any():
                assert np.abs(dev.z[ok] - ref.z[ok]).max() < 1e-9 and np.abs(dev.u0[ok] - ref.u0[ok]).max() < 1e-9


def test_block_layout_prefix_download_and_relayout(track):
    """Inputs / outputs of a batch live in one device block each, laid out for the uploaded batch size: a prefix
    of the batch can still be launched and downloaded, a later upload of another size re-lays the blocks out, and
    the optional outputs (y, z) can be left out of a download."""
    sc = scenarios.make(4, track, B=24, N=10)
    h = _handle(track, 10, "stock", 32)
    full = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    # resident form, prefix of the uploaded batch (download of B' < uploaded takes the per-array path)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    h.solve_resident(9)
    h.sync()
    part = h.download(9, want_y=True)
    for a, b in ((part.z, full.z), (part.y, full.y), (part.u0, full.u0), (part.status, full.status),
                 (part.iters, full.iters), (part.resid, full.resid)):
        assert np.array_equal(a, b[:9])
    # the whole batch, without y (one copy of the block up to z)
    h.solve_resident(24)
    h.sync()
    noy = h.download(24, want_y=False)
    assert noy.y is None and np.array_equal(noy.z, full.z) and np.array_equal(noy.status, full.status)
    # another batch size on the same handle: blocks are laid out again
    sub = slice(5, 12)
    small = h.solve(sc.wp_id[sub], sc.x0[sub], sc.cc_prev[sub], sc.lb[sub], sc.ub[sub], want_y=True)
    assert np.array_equal(small.z, full.z[sub]) and np.array_equal(small.y, full.y[sub])
    assert np.array_equal(small.status, full.status[sub]) and np.array_equal(small.u0, full.u0[sub])
    # and back
    again = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    assert np.array_equal(again.z, full.z) and np.array_equal(again.y, full.y)
    h.close()


def test_errors_are_loud(track):
    cfg = T.stock_config(30, max_batch=4)
    h = mpmpc.Handle(cfg)
    with pytest.raises(mpmpc.MpmpcError):          # no path yet
        h.solve(np.zeros(1, np.int32), np.zeros((1, 3)), np.zeros((1, 60)), np.zeros((1, 30)), np.zeros((1, 30)))
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    with pytest.raises(mpmpc.MpmpcError):          # batch larger than max_batch
        h.solve(np.zeros(8, np.int32), np.zeros((8, 3)), np.zeros((8, 60)), np.zeros((8, 30)), np.zeros((8, 30)))
    with pytest.raises(mpmpc.MpmpcError):          # waypoint out of range
        h.solve(np.array([999], np.int32), np.zeros((1, 3)), np.zeros((1, 60)), np.zeros((1, 30)), np.zeros((1, 30)))
    for bad in (3, -1):                             # tail-kernel modes are 0, 1, 2
        with pytest.raises(mpmpc.MpmpcError):
            h.set_tail_kernel(bad)
    with pytest.raises(mpmpc.MpmpcError):          # K1 timing entry: the batch must have been uploaded
        h.assemble_timed(4, 2)
    h.close()
    with pytest.raises(ValueError):
        T.stock_config(2)


# ---------------------------------------------------------------------------------------------
# host classes on the real library: the reference's drop-in surface
# ---------------------------------------------------------------------------------------------
def _world():
    import test_host_mpc as H
    return H.build_world()


def test_mpc_get_control_replays_reference_lap_on_gpu():
    """MPC.get_control() through libmpmpc.so on the reference's own closed-loop trace (golden G6)."""
    import test_host_mpc as H
    from spatial_bicycle_models import TemporalState
    g = np.load(M.GOLDEN + "/g6_closed_loop_N30.npz")
    m, rp, car = H.build_world()
    # (G6 is the lap of the CERTIFIED stand-in - every proven infeasibility reported: phase1_accept = 0, as in the CPU twin
    #  tests/test_host_mpc.py; the default verdicts are pinned by the stock lap G6s below)
    mpc = H.make_mpc(car, 30, settings=mpmpc.default_settings(phase1_accept=0))                       # real backend
    assert isinstance(mpc.optimizer, mpmpc.Handle)
    n_inf = 0
    for t in range(0, g["s"].size, 2):
        car.s = float(g["s"][t])
        car.temporal_state = TemporalState(*g["pose"][t])
        mpc.current_control = g["cc_prev"][t].copy()
        mpc.infeasibility_counter = int(g["counter"][t - 1]) if t > 0 else 0
        u = mpc.get_control()
        assert car.wp_id == g["wp_id"][t] and mpc.infeasibility_counter == g["counter"][t]
        assert (mpc.last_status > 0) == (g["status"][t] > 0)
        assert np.max(np.abs(u - g["u"][t])) <= 1e-6
        if mpc.last_status > 0:                    # MPC.update_prediction (src/MPC.py:224-248) vs the reference's own
            px, py = mpc.current_prediction
            assert np.max(np.abs(np.array(px) - g["pred_x"][t])) <= 1e-6 and np.max(np.abs(np.array(py) - g["pred_y"][t])) <= 1e-6
        n_inf += int(mpc.last_status < 0)
    assert n_inf > 0


def test_free_running_lap_on_gpu():
    """src/simulation.py's while-loop (simulation.py:134-148) with our classes, plotting off.  With the verdict semantics of
    the solver golden G6 was driven by (every proven infeasibility reported) the free-running lap has the reference lap's
    length to the step; with the default verdicts (marginal instances come back as plans, like the reference's OSQP call) the
    car takes other branches at the marginal steps and the lap may differ by a few steps."""
    import test_host_mpc as H
    g = np.load(M.GOLDEN + "/g6_closed_loop_N30.npz")
    for settings, slack in ((mpmpc.default_settings(phase1_accept=0), 0), (None, 10)):
        m, rp, car = H.build_world()
        mpc = H.make_mpc(car, 30, settings=settings)
        steps = 0
        while car.s < rp.length and steps < 400:
            u = mpc.get_control()
            car.drive(u)
            steps += 1
        assert car.s >= rp.length
        assert abs(steps - g["s"].size) <= slack, (steps, g["s"].size)           # the reference's lap: 210 steps


def test_device_corridor_in_the_single_car_loop():
    """MPC(..., corridor="device"): the corridor of every step comes from the K0 table on the GPU instead of
    ReferencePath.update_path_constraints on the host.  Same controls as the host-corridor controller, step by
    step on identical states, before and after an obstacle is added to the map (the table is rebuilt)."""
    import time
    import test_host_mpc as H
    from map import Obstacle
    m, rp, car = H.build_world()
    host, dev = H.make_mpc(car, 30), H.make_mpc(car, 30, corridor="device")
    t_host = t_dev = 0.0
    for step in range(60):
        if step == 30:
            m.add_obstacles([Obstacle(cx=float(rp.waypoints[car.wp_id + 12].x), cy=float(rp.waypoints[car.wp_id + 12].y), radius=0.03)])
        dev.current_control = np.array(host.current_control)
        dev.infeasibility_counter = host.infeasibility_counter
        t = time.perf_counter(); u_dev = dev.get_control(); t_dev += time.perf_counter() - t
        t = time.perf_counter(); u_host = host.get_control(); t_host += time.perf_counter() - t
        assert dev.last_status == host.last_status
        assert np.max(np.abs(u_dev - u_host)) <= 1e-8, step
        car.drive(u_host)
    assert t_dev < t_host


@pytest.mark.parametrize("cfgid", [2, 3, 4])
def test_device_against_the_independent_leg_and_uniqueness(cfgid, track):
    """VERDICT r2 item 3, on the device.  (a) The uniqueness certificate (oracle/independent.py: null([P; A_S]) vanishes on
    the compared coordinates - plain numpy, no solver) on the DEVICE's own (z, y) of 256 instances of configs 2 / 3 / 4: a
    KKT point of this positive SEMI-definite QP is then THE optimum where it is compared; the count that fails is asserted,
    not excused.  (b) The first 128 instances against golden G8 - restated OSQP ADMM to 1e-10 + ONE stock polish, no code
    shared with the device algorithm, never touched by a device commit: first control and plan to 1e-6."""
    B = 256
    sc = scenarios.make(cfgid, track, B=B)
    h = _handle(track, sc.N, sc.weights, B)
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    h.close()
    solved = np.flatnonzero(sol.status == 1)
    good, n, worst = T.uniqueness_count(qp, sc.N, sol.z, sol.y, solved)
    print("config %d: %d of %d certified optima unique on the compared coordinates (worst freedom %.1e)" % (cfgid, good, n, worst))
    assert n >= 200 and good == n, (good, n, worst)
    g = T.g8("cfg%d" % cfgid)
    sc8 = scenarios.make(cfgid, track, B=int(g["instances"][0]))          # (the seeded batch G8 was made from: a batch's draws depend on its size)
    h = _handle(track, sc8.N, sc8.weights, sc8.B)
    sol8 = h.solve(sc8.wp_id, sc8.x0, sc8.cc_prev, sc8.lb, sc8.ub)
    h.close()
    r = T.compare_with_independent(sol8, g, sc8.N)
    print("config %d against G8:" % cfgid, r)
    assert r["compared"] >= (100 if cfgid != 3 else 120), r
    assert r["worst_u0"] <= 1e-6 and r["worst_plan"] <= 1e-6, r
    assert r["refused_by_device_only"] == 0, r


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_device_against_the_independent_leg_on_the_reference_captures(N, track):
    """The reference's own captured inputs (G4) through libmpmpc.so against golden G8 (independent leg) on the QPs the
    reference assembled from them: same refusals, first control and plan to 1e-6 where G8 certified its own point."""
    g4 = np.load(M.GOLDEN + "/g4_assembly_N%d.npz" % N)
    g = T.g8("g4_N%d" % N)
    h = _handle(track, N, str(g4["weights"][0]), g4["s"].size)
    sol = h.solve(g4["wp_id"].astype(np.int32), g4["x0"], g4["cc_prev"], g4["lb"], g4["ub"])
    h.close()
    r = T.compare_with_independent(sol, g, N)
    assert r["compared"] >= (20 if N != 50 else 52), r
    assert r["worst_u0"] <= 1e-6 and r["worst_plan"] <= 1e-6, r
    assert r["refused_by_device_only"] == 0 and r["refused_by_independent_only"] == 0, r


@pytest.mark.parametrize("cfgid", [4, 5])
def test_default_branch_agreement_with_restated_stock_osqp_on_the_batch_configs(cfgid):
    """VERDICT r4 item 3.  Which branch of src/MPC.py:185-216 get_control takes with the DEFAULT settings - against the
    restated stock OSQP on the full obstacle batch of config 4 and a shard of config 5 (8 192 instances each).  Since round 5
    phase 1 measures infeasibility the way OSQP's ADMM iteration does (Solver::phase1); what is left is pinned here in SIZE,
    DIRECTION and CAUSE: at most 8 instances, every one refused by the device and accepted by OSQP, and every one either
    abandoned by OSQP at max_iter (it then returns its iterate, "solved inaccurate": nothing but its 4 000 iterations can know
    that) or within 0.5 % of OSQP's primal tolerance (OSQP stops an iterate short of its limit point).  Round 4: 28 / 24."""
    r = T.branch_compare(cfgid, 8192)
    assert r["agreement"] >= 0.999 and len(r["rows"]) <= 8, r
    for i, dev_status, dev_viol, st_status, st_iters, st_pri in r["rows"]:
        assert dev_status == mpmpc.PRIMAL_INFEASIBLE and st_status in (1, 2), (i, dev_status, st_status)
        assert st_iters >= 4000 or abs(dev_viol / r["threshold"] - 1.0) <= 5e-3, (i, dev_viol, st_iters, st_pri)
        assert dev_viol > r["threshold"] * (1 - 1e-9)          # the device's own rule: refused means beyond OSQP's tolerance
    assert r["device"].get(2, 0) >= 60 and r["device"].get(-3, 0) >= 600


@pytest.mark.parametrize("N", [10, 30])
def test_default_path_takes_the_branch_stock_osqp_takes_on_device(N, track):
    """ADVICE r2 (high), on the device: golden G6s is the reference's own loop run with the restated OSQP at ITS DEFAULTS
    (eps 1e-3, no polish, no phase 1 - the arithmetic of src/MPC.py:159,183).  Every recorded step through libmpmpc.so
    with the DEFAULT settings takes the branch the reference took: a usable plan (status 1 or 2) exactly where OSQP
    returned one, a refusal exactly where OSQP refused.  With phase1_accept = 0 the same batch refuses the marginal
    steps too - the behaviour ADVICE r2 objected to."""
    g = np.load(M.GOLDEN + "/g6s_stock_loop_N%d.npz" % N)
    B = g["s"].size
    h = _handle(track, N, "stock", B)
    sol = h.solve(g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"], want_y=True)
    qp = h.assemble(g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"])
    h.close()
    usable = (sol.status == 1) | (sol.status == 2)
    assert np.array_equal(usable, g["status"] > 0), np.flatnonzero(usable != (g["status"] > 0))
    assert (sol.status == 2).sum() >= 5
    # what a status-2 plan IS, checked without the device's code (VERDICT r3 "weak" 3): the unique KKT point of the QP whose
    # boxes are relaxed to what the plan uses, none of them by more than 1.5 x the violation reported in resid[0]
    for i in np.flatnonzero(sol.status == 2):
        c = T.relaxed_plan_check(qp[:, i, :], N, sol.z[i], sol.y[i], sol.resid[i, 0])
        assert c["kkt"] <= 1e-8 and c["unique"] and 0.0 < c["relaxation"] <= 1.5 * sol.resid[i, 0] * (1 + 1e-6) + 1e-12, (i, c)
    assert np.max(np.abs(sol.u0[usable, 0] - g["u"][usable, 0])) <= 5e-3          # speed channel: OSQP at 1e-3 has it
    hs = _handle(track, N, "stock", B, mpmpc.default_settings(**STRICT))
    strict = hs.solve(g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"])
    hs.close()
    assert (strict.status == -3).sum() >= (sol.status == -3).sum() + 5


def test_batch_mpc_matches_single_controller():
    import test_host_mpc as H
    from MPC import BatchMPC
    from scipy import sparse
    m, rp, car = H.build_world()
    g = np.load(M.GOLDEN + "/g6_closed_loop_N30.npz")
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    sc = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    idx = np.arange(0, 200, 5)
    bm = BatchMPC(car, 30, Q, R, QN, sc, ic, 4.0, max_batch=idx.size, settings=mpmpc.default_settings(**STRICT))
    wp, x0 = bm.spatial_states(g["s"][idx], g["pose"][idx])
    assert np.array_equal(wp, g["wp_id"][idx]) and np.allclose(x0, g["x0"][idx], atol=1e-13)
    u, plan, status, sol = bm.get_control_batch(wp, x0, g["cc_prev"][idx], g["lb"][idx], g["ub"][idx])
    ok = g["status"][idx] > 0
    assert np.array_equal(status > 0, ok)
    assert np.max(np.abs(u[ok] - g["u"][idx][ok])) <= 1e-6
    d = np.abs(plan[ok] - g["cc_next"][idx][ok])
    d[:, -1] = 0.0
    assert d.max() <= 1e-6


def test_maximum_horizon_and_finite_state_boxes_gpu(track):
    import oracle_c as OC
    N, B = 63, 6
    sc = scenarios.make(2, track, B=B, N=50)
    lb = np.concatenate([sc.lb, np.repeat(sc.lb[:, -1:], N - 50, axis=1)], axis=1)
    ub = np.concatenate([sc.ub, np.repeat(sc.ub[:, -1:], N - 50, axis=1)], axis=1)
    cc = np.zeros((B, 2 * N))
    Q, R, QN = scenarios.WEIGHTS["stock"]
    xmin, xmax = np.array([-np.inf, -0.6, -np.inf]), np.array([np.inf, 0.6, 4.0])
    cfg = mpmpc.make_config(N, Q, R, QN, xmin, xmax, scenarios.UMIN, scenarios.UMAX, 4.0, 0.12, max_batch=B)
    h = mpmpc.Handle(cfg)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    sol = h.solve(sc.wp_id, sc.x0, cc, lb, ub)
    ocfg = OC.mpc_cfg(N, (Q, R, QN), scenarios.UMIN, scenarios.UMAX, xmin, xmax, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, cc, lb, ub)
    assert np.array_equal(sol.status, ref["status"]) and np.all(sol.status == 1)
    assert np.max(np.abs(sol.u0 - ref["u0"])) <= 1e-6
    h.close()


@pytest.mark.parametrize("cfgid,B", [(2, 1024), (3, 4096), (4, 8192)])
def test_full_batches_against_c_oracle(cfgid, B, track):
    """The FULL batches of configs 2, 3 and 4 (incl. the two-instances-per-wave kernel + tail launch at B = 8192)
    against the C port of the oracle: EVERY status, every ADMM iteration count, every control; and every instance
    ends with a certificate - KKT for the solved ones, a Farkas ray for the infeasible ones - nobody runs OSQP's ADMM
    beyond the one iteration of the early attempt."""
    import oracle_c as OC
    sc = scenarios.make(cfgid, track, B=B)
    h = _handle(track, sc.N, sc.weights, B, mpmpc.default_settings(**STRICT))
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN,
                      scenarios.XMAX, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, sc.cc_prev,
                       sc.lb, sc.ub, want_y=True)
    assert np.array_equal(sol.status, ref["status"])
    assert np.array_equal(sol.iters[:, 0], ref["iters"][:, 0])
    both = sol.status == 1
    assert both.mean() > 0.85 and set(np.unique(sol.status)) <= {1, mpmpc.PRIMAL_INFEASIBLE}
    worst, alt = T.controls_vs_reference(qp, sc.N, sol, ref, 1e-6)
    assert worst <= 1e-6, (worst, alt)
    # an instance whose control differs while both points carry a KKT certificate and equal objectives is excused only with
    # a PROOF that its optimum is not a point: the uniqueness certificate must fail on the compared coordinates
    # (VERDICT r3 "weak" 4; config 4 has had one weakly active corridor bound in 8 192 instances)
    import independent as I
    keep, _ = I.compared_coordinates(sc.N)
    for i in alt:
        Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
        assert not I.uniqueness_certificate(Pd, A, l, u, sol.z[i], sol.y[i], keep)["unique"], i
    prim, stat, comp = T.kkt_batch(qp[:, both, :], sc.N, sol.z[both], sol.y[both])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    inf = ~both
    if cfgid == 4:
        assert inf.sum() > 500
    if inf.any():
        ok, support, aty = T.farkas_batch(qp[:, inf, :], sc.N, sol.y[inf])
        assert ok.all(), (support.max(), aty.max())
    h.close()


@pytest.mark.parametrize("cfgid,B", [(4, 2048), (2, 256)])
def test_stock_mode_against_c_oracle(cfgid, B, track):
    """polish = 0 - OSQP's own ADMM at its defaults, what src/MPC.py:159,183 runs - on the obstacle course: the
    infeasibility verdicts, the "solved" calls and the iteration counts of the restated OSQP, instance by instance."""
    import oracle_c as OC
    sc = scenarios.make(cfgid, track, B=B)
    h = _handle(track, sc.N, sc.weights, B, mpmpc.default_settings(polish=0, early_polish=0))
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN,
                      scenarios.XMAX, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(polish=0, early_polish=0), track.kappa, track.v_ref, track.ds_next, sc.wp_id,
                       sc.x0, sc.cc_prev, sc.lb, sc.ub)
    assert np.array_equal(sol.status, ref["status"])
    assert np.array_equal(sol.iters[:, 0], ref["iters"][:, 0])
    if cfgid == 4:
        assert (sol.status == mpmpc.PRIMAL_INFEASIBLE).sum() > 100 and sol.iters[:, 0].max() >= 1000
    ok = sol.status == 1
    assert np.max(np.abs(sol.z[ok] - ref["z"][ok])) <= 1e-6
    h.close()


def test_default_path_does_not_depend_on_adaptive_rho_interval(track):
    """Stock OSQP derives adaptive_rho_interval from wall-clock time (not reproducible; its deterministic fallback is
    4 x check_termination = 100, the build's fixed value is 50).  The default path never gets as far as a rho update
    - every instance is certified optimal or certified infeasible after ONE ADMM iteration - so its outputs are
    bit-identical for 25 / 50 / 100.  (The stock mode, polish = 0, does depend on it: reported, not asserted equal.)"""
    sc = scenarios.make(4, track, B=2048)
    outs = []
    for interval in (25, 50, 100):
        h = _handle(track, sc.N, sc.weights, sc.B, mpmpc.default_settings(adaptive_rho_interval=interval))
        outs.append(h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
        h.close()
    for o in outs[1:]:
        assert np.array_equal(o.status, outs[0].status) and np.array_equal(o.z, outs[0].z) and np.array_equal(o.iters, outs[0].iters)
    stock = []
    for interval in (25, 50, 100):
        h = _handle(track, sc.N, sc.weights, 512, mpmpc.default_settings(polish=0, early_polish=0, adaptive_rho_interval=interval))
        stock.append(h.solve(sc.wp_id[:512], sc.x0[:512], sc.cc_prev[:512], sc.lb[:512], sc.ub[:512]))
        h.close()
    # the verdict classes that drive get_control's branch (usable plan or not) agree between the intervals
    usable = [np.isin(o.status, (1, 2)) for o in stock]
    print("stock mode, adaptive_rho_interval 25/50/100: usable-plan verdicts differing from 50:",
          int((usable[0] != usable[1]).sum()), int((usable[2] != usable[1]).sum()), "of 512")
    assert np.mean(usable[0] == usable[1]) >= 0.99 and np.mean(usable[2] == usable[1]) >= 0.99


@pytest.mark.parametrize("cfgid,B", [(2, 512), (4, 1024), (3, 256)])
def test_the_polish_settings_change_the_route_not_the_answer(cfgid, B, track):
    """The knobs of the certified polish - centred or warm start, step indicators' tolerance, selective active-set
    additions, reduced or full problem - decide how many iterations an instance takes, never what it returns: every
    variant ends with the same status and, where solved, the same certified optimum (1e-8 in u; both certified)."""
    sc = scenarios.make(cfgid, track, B=B)
    variants = [dict(), dict(ipm_start_mu=0.0), dict(ipm_tol=1e-9), dict(ipm_tol=1e-7), dict(as_add_fraction=0.0),
                dict(ipm_start_slack=0.3, ipm_start_mu=0.1), dict(ipm_start_dual=0.0), dict(reduce=0), dict(native=0),
                dict(native_ipm_tol=1e-8), dict(native_ipm_tol=1e-6)]
    outs = []
    for kw in variants:
        h = _handle(track, sc.N, sc.weights, B, mpmpc.default_settings(**STRICT, **kw))
        if not outs:
            qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        outs.append(h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True))
        h.close()
    ref = outs[0]
    ok = ref.status == 1
    assert set(np.unique(ref.status)) <= {1, mpmpc.PRIMAL_INFEASIBLE} and ok.mean() > 0.85
    for kw, o in zip(variants[1:], outs[1:]):
        assert np.array_equal(o.status, ref.status), kw
        worst, alt = T.controls_vs_reference(qp, sc.N, o, dict(status=ref.status, u0=ref.u0, z=ref.z, y=ref.y), 1e-8)
        assert worst <= 1e-8 and alt.size <= 1, (kw, worst, alt)
        # (an instance excused as an "alternative optimum" needs a PROOF that its optimum is not a point: the uniqueness
        #  certificate must FAIL on the compared coordinates - as in test_full_batches_against_c_oracle; VERDICT r4 "weak" 10)
        if alt.size:
            import independent as I
            keep, _ = I.compared_coordinates(sc.N)
            for i in alt:
                Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
                assert not I.uniqueness_certificate(Pd, A, l, u, o.z[i], o.y[i], keep)["unique"], (kw, i)
        prim, stat, comp = T.kkt_batch(qp[:, ok, :], sc.N, o.z[ok], o.y[ok])
        assert max(prim.max(), stat.max(), comp.max()) <= 1e-8, kw


@pytest.mark.parametrize("cfgid,B", [(5, 65536), (2, 65536)])
def test_baseline_batch_sizes_carry_kkt_certificates(cfgid, B, track):
    """BASELINE.json's largest batches (config 5: 65 536 obstacle-course instances; config 2's poses at
    the same size) through a size-independent property: every instance reported solved satisfies the
    KKT conditions of ITS OWN assembled QP to 1e-8 (vectorised numpy on K1's output and K2's z, y),
    every instance of the obstacle course is either certified optimal or carries a Farkas ray that proves it infeasible,
    and the batch is invariant (statuses, iteration counts; controls to 1e-9) under a permutation of the
    instances: no cross-talk between the instances that share a wavefront."""
    sc = scenarios.make(cfgid, track, B=B)
    h = _handle(track, sc.N, sc.weights, B)
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    ok = sol.status == 1
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], sc.N, sol.z[ok], sol.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    # nobody needs OSQP's ADMM beyond the one iteration of the early attempt: every instance is certified optimal or
    # proved infeasible - including the fifteen of config 5 whose corridor cannot be met by a tenth of a millimetre
    # (phase 1's converged optimum, criterion B)
    # ... or, infeasible by less than OSQP's own primal tolerance, handed back as a usable plan (status 2, phase1_accept)
    assert set(np.unique(sol.status)) <= {1, 2, -3}
    assert np.all(sol.iters[:, 0] == 1)
    if cfgid == 2:
        assert ok.all()
    else:
        assert 0.85 < ok.mean() < 0.95 and (sol.status == -3).mean() > 0.03
        inf = sol.status == -3
        fk, support, aty = T.farkas_batch(qp[:, inf, :], sc.N, sol.y[inf])
        assert fk.all(), (support.max(), aty.max())
        # the marginal ones: the plan keeps the dynamics exactly and leaves its boxes by at most 1.5 x the least violation
        # reported in resid[0], which is below OSQP's tolerance eps_abs + eps_rel |kappa_max|; the strict setting
        # reports exactly these instances (and the -3 ones) infeasible
        mg = sol.status == 2
        assert 0 < mg.sum() < 0.06 * B
        eps_osqp = 1e-3 + 1e-3 * float(np.max(np.abs(scenarios.UMAX)))
        assert sol.resid[mg, 0].max() <= eps_osqp and sol.resid[mg, 0].min() > 1e-8
        prim_m, _, _ = T.kkt_batch(qp[:, mg, :], sc.N, sol.z[mg], sol.y[mg])
        assert np.all(prim_m <= 1.5 * sol.resid[mg, 0] + 1e-8) and np.all(prim_m > 1e-7)
        hs = _handle(track, sc.N, sc.weights, B, mpmpc.default_settings(**STRICT))
        strict = hs.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        hs.close()
        assert np.array_equal(strict.status == -3, mg | inf) and np.array_equal(strict.status == 1, ok)
        assert np.array_equal(strict.u0[ok], sol.u0[ok])
    # ... and the C oracle on the WHOLE batch (VERDICT r4 "weak" 9: it does 65 536 instances in seconds): every status of the
    # certified port (the device's strict verdicts), every control to 1e-6
    import oracle_c as OC
    ocfg = OC.mpc_cfg(sc.N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN, scenarios.XMAX, 4.0, 0.12)
    ref = OC.mpc_batch(ocfg, OC.settings(), track.kappa, track.v_ref, track.ds_next, sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    assert np.array_equal(ref["status"] == 1, ok) and np.array_equal(ref["status"] == -3, sol.status != 1)
    worst, alt = T.controls_vs_reference(qp, sc.N, sol, ref, 1e-6)
    assert worst <= 1e-6 and alt.size <= 4, (worst, alt)
    if alt.size:
        import independent as I
        keep, _ = I.compared_coordinates(sc.N)
        for i in alt:
            # (every optimal point pairs with the multipliers of ANY KKT point: the proof may use either side's.  A weakly
            #  active corridor bound carries a multiplier of 1e-6 on one side and exactly zero on the other - 65 536 instances
            #  hold a handful of those)
            Pd, q, A, l, u = T.qp_to_dense(qp[:, i, :], sc.N)
            ud = I.uniqueness_certificate(Pd, A, l, u, sol.z[i], sol.y[i], keep)
            ur = I.uniqueness_certificate(Pd, A, l, u, ref["z"][i], ref["y"][i], keep)
            if ud["unique"] and ur["unique"]:
                # ... or the optimum IS a point, but in a direction so flat (smallest singular value of [P; A_S] below 1e-3:
                # curvature 1e-7) that two points 6e-5 apart both pass a 1e-8 certificate with objectives equal to 1e-15
                # (instance 15 869 of config 5: delta_0 differs by 2.7e-6).  Then the DEVICE's point must be the sharper of the
                # two: its KKT residuals at rounding level (measured 2e-16; the C port's complementarity 6e-10, from the 1e-9
                # regularisation its general LDL needs)
                kd = np.max(T.kkt_batch(qp[:, [i], :], sc.N, sol.z[[i]], sol.y[[i]]))
                kr = np.max(T.kkt_batch(qp[:, [i], :], sc.N, ref["z"][[i]], ref["y"][[i]]))
                assert ud["smallest_sv"] < 1e-3 and kd <= 1e-12 and kd <= kr, (i, ud["smallest_sv"], kd, kr)
    perm = np.random.default_rng(0).permutation(B)
    sol2 = h.solve(sc.wp_id[perm], sc.x0[perm], sc.cc_prev[perm], sc.lb[perm], sc.ub[perm])
    assert np.array_equal(sol2.status, sol.status[perm]) and np.array_equal(sol2.iters, sol.iters[perm])
    # (not bit-for-bit: instances sharing a wavefront run the refinement loops until both are done)
    both = sol.status[perm] == 1
    assert np.max(np.abs(sol2.u0[both] - sol.u0[perm][both])) <= 1e-9
    h.close()


@pytest.mark.parametrize("cfgid,N", [(2, 10), (4, 10), (4, 3), (2, 15)])
def test_every_lane_packing_gives_the_same_answers(cfgid, N, track):
    """64, 32 and 16 lanes per instance (1, 2, 4 instances per wavefront; the launcher picks by batch
    size, mpmpc_set_packing forces one): same statuses, same iteration counts, same controls."""
    sc = scenarios.make(cfgid, track, B=203, N=N)
    h = _handle(track, sc.N, sc.weights, sc.B)
    sols = {}
    for g in (64, 32, 16):
        h.set_packing(g)
        sols[g] = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    h.set_packing(0)
    for g in (32, 16):
        assert np.array_equal(sols[g].status, sols[64].status) and np.array_equal(sols[g].iters[:, 0], sols[64].iters[:, 0])
        ok = sols[64].status == 1
        assert np.max(np.abs(sols[g].u0[ok] - sols[64].u0[ok])) <= 1e-9
    h.close()


def test_non_finite_inputs_get_no_verdict_on_device(track):
    """NaN / Inf poses: status UNSOLVED (-10), never a solved plan, never a hang; the other instances of
    the batch (and of the same wavefront, with 32 lanes per instance) are untouched."""
    sc = scenarios.make(2, track, B=6)
    h = _handle(track, sc.N, sc.weights, sc.B)
    x0 = sc.x0.copy()
    x0[1, 0], x0[2, 1] = np.nan, np.inf
    clean = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for g in (64, 32):
        h.set_packing(g)
        sol = h.solve(sc.wp_id, x0, sc.cc_prev, sc.lb, sc.ub)
        assert list(sol.status[[1, 2]]) == [mpmpc.UNSOLVED, mpmpc.UNSOLVED]
        for i in (0, 3, 4, 5):
            assert sol.status[i] == clean.status[i] == 1 and np.max(np.abs(sol.u0[i] - clean.u0[i])) <= 1e-9
    h.close()


def test_open_path_end_is_an_error(track):
    cfg = T.stock_config(30, max_batch=2, circular=False)
    h = mpmpc.Handle(cfg)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    with pytest.raises(mpmpc.MpmpcError, match="Reached end of path"):
        h.solve(np.array([180], np.int32), np.zeros((1, 3)), np.zeros((1, 60)), np.zeros((1, 30)), np.zeros((1, 30)))
    ok = h.solve(np.array([100], np.int32), np.zeros((1, 3)), np.zeros((1, 60)), np.full((1, 30), -0.1),
                 np.full((1, 30), 0.1))
    assert ok.status[0] == 1
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 5, 63, 64, 65, 128, 129, 199, 300, 420])
def test_speed_profile_path_lengths(n):
    """K4's wave-per-path kernel (lane-strided loops, cyclic reduction of the tridiagonal systems) for path lengths
    around the wave width and the powers of two, and - n = 420 - the thread-per-path kernel that takes over when
    the work arrays no longer fit the LDS: all against the emulation of the scalar code."""
    rng = np.random.default_rng(100 + n)
    B = 7
    LI = rng.uniform(0.03, 0.06, (B, n))
    KA = rng.normal(0.0, 2.0, (B, n)) * (rng.uniform(0, 1, (B, n)) < 0.5)
    LIM = np.stack([-rng.uniform(0.05, 0.5, B), rng.uniform(0.1, 1.0, B), np.zeros(B), rng.uniform(0.6, 1.5, B),
                    rng.uniform(1.0, 5.0, B)], axis=1)
    v, status, iters = mpmpc.speed_profile(LI, KA, LIM)
    ve, se, _ = T.emu_speed_profile(LI, KA, LIM)
    assert np.array_equal(status, se) and np.all(status == 1)
    assert np.max(np.abs(v - ve)) < 1e-9


def test_speed_profile_on_device():
    """K4 through the C ABI: the reference track's profile against G2 (the certified optimum of the
    QP the reference hands to OSQP, src/reference_path.py:289-354), and a batch of perturbed paths
    against the emulation of the same code."""
    G = M.GOLDEN
    g1 = np.load(G + "/g1_path_sim_track.npz")
    g2 = np.load(G + "/g2_speed_profile.npz")
    n = 199
    li, kappa = g1["ds_next"][:n], g1["kappa"][:n].astype(float)
    v, status, iters = mpmpc.speed_profile(li, kappa, g2["constraints"])
    assert status[0] == 1 and 3 < iters[0] < 40
    assert np.max(np.abs(v[0] - g2["x"])) < 1e-9
    rng = np.random.default_rng(12)
    B = 96                                                # LDS variant; the HBM-workspace variant below
    LI = li[None, :] * rng.uniform(0.7, 1.4, (B, n))
    KA = kappa[None, :] * rng.uniform(0.5, 3.0, (B, 1))
    LIM = np.stack([-rng.uniform(0.05, 0.5, B), rng.uniform(0.1, 1.0, B), np.zeros(B), rng.uniform(0.6, 1.5, B),
                    rng.uniform(1.0, 5.0, B)], axis=1)
    LIM[5] = [0.5, -0.1, 0.0, 1.0, 4.0]                     # inconsistent: refused
    v, status, iters = mpmpc.speed_profile(LI, KA, LIM)
    ve, se, _ = T.emu_speed_profile(LI, KA, LIM)
    assert np.array_equal(status, se) and status[5] == -1 and np.all(np.delete(status, 5) == 1)
    ok = status == 1
    assert np.max(np.abs(v[ok] - ve[ok])) < 1e-9
    Bbig = 384
    rep = np.arange(Bbig) % B
    vb, sb, _ = mpmpc.speed_profile(LI[rep], KA[rep], LIM[rep])
    assert np.array_equal(sb, status[rep]) and np.array_equal(vb[sb == 1], v[rep][sb == 1])
    # the host class takes its profile from the device
    from map import Map
    from reference_path import ReferencePath
    h, w = g1["grid_shape"]
    m = Map.from_grid(np.unpackbits(g1["grid_free"])[:h * w].reshape(h, w).astype(np.int8), origin=[-1, -2], resolution=0.005)
    rp = ReferencePath.from_tables(m, g1["x"], g1["y"], g1["psi"], g1["kappa"], circular=True,
                                   border_ub=g1["border_ub"], border_lb=g1["border_lb"])
    rp.compute_speed_profile(dict(zip(('a_min', 'a_max', 'v_min', 'v_max', 'ay_max'), g2["constraints"])))
    assert np.max(np.abs(np.array([w_.v_ref for w_ in rp.waypoints]) - g2["v_ref"])) < 1e-9


@pytest.mark.parametrize("cfgid,N,B", [(2, 30, 48), (4, 30, 1500), (3, 50, 12)])
def test_full_terminal_weight_on_device(cfgid, N, B, track, emu):
    """QN with off-diagonal entries (src/MPC.py:150,154 use the whole matrix) through libmpmpc.so: the device agrees with
    the lock-step emulation of the same lane code (which the CPU suite checks against the numpy restatement of the
    reference's full-QN assembly and the dense oracle), every solved instance passes the KKT certificate on the QP with
    the dense terminal block, and a batch beyond 1024 still runs (one instance per wave: the packed kernels have no
    dense-block code)."""
    from test_emul_parity import QN_FULL, _dense_with_qn
    sc = scenarios.make(cfgid, track, B=B, N=N)
    Q, R, _ = scenarios.WEIGHTS[sc.weights]
    cfg = mpmpc.make_config(N, Q, R, QN_FULL, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                            scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=B)
    h = mpmpc.Handle(cfg)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    h.close()
    n = min(B, 96)
    ref = emu.solve(cfg, mpmpc.default_settings(), np.ascontiguousarray(qp[:, :n, :]), G=64)
    assert np.array_equal(sol.status[:n], ref.status) and np.array_equal(sol.iters[:n, 0], ref.iters[:, 0])
    ok = ref.status == 1
    assert ok.sum() >= n // 2 and np.max(np.abs(sol.z[:n][ok] - ref.z[ok])) <= 1e-9
    for i in np.flatnonzero(ok)[:24]:
        P, q, A, l, u = _dense_with_qn(qp[:, i, :], N, QN_FULL)
        assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8)
    assert set(np.unique(sol.status)) <= {1, 2, -3}


@pytest.mark.parametrize("N", [3, 10, 30, 50])
def test_full_stage_weights_on_device_reference_captures(N, track, emu):
    """non-diagonal Q and R through the drop-in.  Golden G4f = what the REFERENCE handed to osqp.setup with
    non-diagonal Q, R, QN.  Through libmpmpc.so: K1's fields + the configuration's off-diagonals rebuild the reference's
    (P, q, A, l) bit for bit (u to the ulps of the device's tan), the general kernel (dense 3 x 3 / 2 x 2 stage blocks) agrees
    with its lock-step emulation, reaches the C oracle's optimum of the CAPTURED QP (statuses, z to 1e-6, first control to
    1e-8) and every solved instance carries the plain-numpy KKT certificate and the uniqueness certificate on the captured data."""
    from scipy import sparse
    import oracle_c as OC
    import independent as I
    from test_emul_parity import full_weight_config
    g = np.load(M.GOLDEN + "/g4f_assembly_N%d.npz" % N)
    B, n, m = g["s"].size, 5 * N + 3, 8 * N + 6
    cfg = full_weight_config(N, "full", max_batch=B)
    h = mpmpc.Handle(cfg)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    args = (g["wp_id"].astype(np.int32), g["x0"], g["cc_prev"], g["lb"], g["ub"])
    qp = h.assemble(*args)
    sol = h.solve(*args, want_y=True)
    h.close()
    ref = emu.solve(cfg, mpmpc.default_settings(), qp, G=64)
    assert np.array_equal(sol.status, ref.status) and np.array_equal(sol.iters[:, 0], ref.iters[:, 0])
    ok = ref.status == 1
    assert np.max(np.abs(sol.z[ok] - ref.z[ok])) <= 1e-9
    keep, _ = I.compared_coordinates(N)
    n_ok = 0
    for c in range(B):
        P, q, A, l, u = T.qp_to_dense_full(qp[:, c, :], N, cfg)
        Pref = sparse.coo_matrix((g["P_val"][c], (g["P_row"][c], g["P_col"][c])), shape=(n, n)).toarray()
        lo, hi = g["A_case_ptr"][c], g["A_case_ptr"][c + 1]
        Aref = sparse.csc_matrix((g["A_data"][lo:hi], g["A_indices"][lo:hi], g["A_indptr"][c]), shape=(m, n)).toarray()
        assert np.array_equal(P, Pref) and np.array_equal(q, g["q"][c]) and np.array_equal(A, Aref)
        assert np.array_equal(l, np.where(g["l"][c] <= -1e30, -np.inf, g["l"][c]))
        fin = np.isfinite(g["u"][c])
        assert np.max(np.abs(u[fin] - g["u"][c][fin])) <= 8 * np.finfo(float).eps
        x, y, info = OC.solve(Pref, g["q"][c], Aref, g["l"][c], g["u"][c])
        assert sol.status[c] == info.status or (sol.status[c] == 2 and info.status == -3), (c, sol.status[c], info.status)
        if info.status == 1 and info.polished == 1:
            n_ok += 1
            e = np.abs(sol.z[c] - x)
            e[-1] = 0.0
            assert e.max() <= 1e-6 and np.max(e[3 * (N + 1):3 * (N + 1) + 2]) <= 1e-8
            assert O.kkt_certificate(Pref, g["q"][c], Aref, g["l"][c], g["u"][c], sol.z[c], sol.y[c])["ok_tol"](1e-8)
            assert I.uniqueness_certificate(Pref, Aref, g["l"][c], g["u"][c], sol.z[c], sol.y[c], keep)["unique"], c
    assert n_ok >= B // 2


@pytest.mark.parametrize("name,cfgid,N,B", [("full", 4, 30, 1500), ("rank1", 2, 30, 200), ("q_only", 4, 10, 300), ("r_only", 3, 50, 100)])
def test_full_stage_weights_on_device_batches(name, cfgid, N, B, track, emu):
    """Non-diagonal stage weights on BASELINE-type batches (beyond 1 024 instances too: one instance per wave, the packed
    kernels carry no dense-block code): device = emulation of the same lane code, KKT certificate with plain numpy on the dense
    data for every solved instance checked, Farkas rays for the refused ones; singular (rank-one) weight blocks included."""
    from test_emul_parity import full_weight_config, FULL_WEIGHT_SETS
    sc = scenarios.make(cfgid, track, B=B, N=N)
    cfg = full_weight_config(N, name, max_batch=B)
    h = mpmpc.Handle(cfg, mpmpc.default_settings(**STRICT))
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    sol = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    h.close()
    nn = min(B, 64)
    ref = emu.solve(cfg, mpmpc.default_settings(**STRICT), np.ascontiguousarray(qp[:, :nn, :]), G=64)
    assert np.array_equal(sol.status[:nn], ref.status) and np.array_equal(sol.iters[:nn], ref.iters)
    ok = ref.status == 1
    f = lambda P, q, x: 0.5 * x @ P @ x + q @ x
    if name != "rank1":
        assert np.max(np.abs(sol.z[:nn][ok] - ref.z[ok])) <= 1e-9
    assert set(np.unique(sol.status)) <= {1, -3} and (sol.status == 1).mean() > 0.8
    rng = np.random.default_rng(0)
    for i in rng.permutation(B)[:40]:
        P, q, A, l, u = T.qp_to_dense_full(qp[:, i, :], N, cfg)
        if sol.status[i] == 1:
            assert O.kkt_certificate(P, q, A, l, u, sol.z[i], sol.y[i])["ok_tol"](1e-8), i
        else:
            assert O.farkas_certificate(A, l, u, sol.y[i], 1e-6)["ok"], i
    for i in np.flatnonzero(ok)[:8]:             # objective against the emulation (flat directions of singular blocks excused)
        P, q, A, l, u = T.qp_to_dense_full(qp[:, i, :], N, cfg)
        assert abs(f(P, q, sol.z[i]) - f(P, q, ref.z[i])) <= 1e-10 * max(1.0, abs(f(P, q, ref.z[i])))


def test_diagonal_weights_given_as_matrices_keep_the_reduced_native_kernels(track):
    """Diagonal Q, R, QN handed over as full MATRICES (dense or scipy sparse, as src/simulation.py:101-103 does) give the very
    configuration the vectors give - byte for byte - and therefore the same dispatch (reduced-native kernels) and the same
    bits; one non-zero off-diagonal entry moves the configuration to the general kernels (same statuses, another optimum)."""
    from scipy import sparse
    sc = scenarios.make(4, track, B=600)
    Q, R, QN = scenarios.WEIGHTS["stock"]
    mk = lambda q, r, qn: mpmpc.make_config(sc.N, q, r, qn, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                                            scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=sc.B)
    c_vec, c_mat, c_sp = mk(Q, R, QN), mk(np.diag(Q), np.diag(R), np.diag(QN)), mk(sparse.diags(Q), sparse.diags(R), sparse.diags(QN))
    assert bytes(c_vec) == bytes(c_mat) == bytes(c_sp)
    sols = []
    for c in (c_vec, c_mat):
        h = mpmpc.Handle(c)
        h.set_path(track.kappa, track.v_ref, track.ds_next)
        sols.append(h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
        h.close()
    assert np.array_equal(sols[0].z, sols[1].z) and np.array_equal(sols[0].status, sols[1].status)
    Rf = np.diag(R).copy()
    Rf[0, 1] = Rf[1, 0] = 1e-3
    Qf = np.diag(Q).copy()
    Qf[1, 1] = 1e-4                      # (R must stay PSD: its kappa weight is zero, so the coupling needs a positive one)
    Rf[1, 1] = 1e-4
    h = mpmpc.Handle(mk(Qf, Rf, QN))
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    s2 = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    h.close()
    both = (sols[0].status == 1) & (s2.status == 1)
    assert both.mean() > 0.8 and np.max(np.abs(s2.u0[both] - sols[0].u0[both])) > 1e-7


def test_paths_longer_than_the_lds_staging_of_k1(emu):
    """K1 stages the three path tables in LDS when the path has at most 1 024 waypoints and reads them from memory
    otherwise (the limit's far side was untested): a 1 500-waypoint synthetic path, assembly bit for bit and
    solves to 1e-9 against the emulation of the same lane code, horizon windows wrapping around the end of the path."""
    rng = np.random.default_rng(5)
    n_wp, N, B = 1500, 30, 96
    kappa = 3.0 * np.sin(np.arange(n_wp) * 0.021) * (rng.uniform(0, 1, n_wp) < 0.8)
    v_ref = rng.uniform(0.6, 1.0, n_wp)
    ds = rng.uniform(0.03, 0.05, n_wp)

    class Tr:
        pass
    tr = Tr()
    tr.kappa, tr.v_ref, tr.ds_next = kappa, v_ref, ds
    tr.ub_free = tr.lb_free = tr.ub_obstacles = tr.lb_obstacles = np.zeros((2, N))
    wp = np.concatenate([rng.integers(0, n_wp, B - 4), [n_wp - 1, n_wp - 2, n_wp - N, 0]]).astype(np.int32)
    x0 = np.stack([rng.uniform(-0.02, 0.02, B), rng.uniform(-0.2, 0.2, B), np.zeros(B)], axis=1)
    cc = np.zeros((B, 2 * N))
    cc[B // 2:, 0::2] = 0.8
    cc[B // 2:, 1::2] = rng.uniform(-0.3, 0.3, (B - B // 2, N))
    lb, ub = np.full((B, N), -0.15) + rng.uniform(0, 0.05, (B, N)), np.full((B, N), 0.15) - rng.uniform(0, 0.05, (B, N))
    cfg = T.stock_config(N, max_batch=B)
    h = mpmpc.Handle(cfg)
    h.set_path(kappa, v_ref, ds)
    qp = h.assemble(wp, x0, cc, lb, ub)
    sol = h.solve(wp, x0, cc, lb, ub)
    h.close()
    qp_e = emu.assemble(cfg, tr, (wp, x0, cc, lb, ub))
    cap = 15                                   # the speed cap goes through tan(): device libm vs the host's, a few ulp
    other = np.delete(np.arange(mpmpc.NUM_FIELDS), cap)
    assert np.array_equal(qp[other][:, :, :N + 1], qp_e[other][:, :, :N + 1])        # (the padding stage is never written)
    assert np.max(np.abs(qp[cap, :, :N + 1] - qp_e[cap, :, :N + 1])) <= 8 * np.finfo(float).eps
    ref, _ = emu.solve_launch(cfg, mpmpc.default_settings(), qp_e, G=64)
    assert np.array_equal(sol.status, ref.status) and np.all(sol.status == 1)
    assert np.max(np.abs(sol.z - ref.z)) <= 1e-9


@pytest.mark.parametrize("cfgid,B", [(2, 1024), (4, 4096)])
def test_reduced_polish_gives_the_full_polish_answers(cfgid, B, track):
    """reduce = 1 (default: interior point / active set / phase 1 on the (e_y, e_psi, kappa) problem, v in closed form, t
    rolled forward) against reduce = 0 (the full 3-state polish) on the same batches: same verdicts, the same optimum
    to 1e-9 in every entry of z that the cost determines, and every point of the reduced path passes the KKT
    certificate of the FULL problem (numpy on K1's output)."""
    sc = scenarios.make(cfgid, track, B=B)
    out = {}
    for red in (1, 0):
        h = _handle(track, sc.N, sc.weights, B, mpmpc.default_settings(reduce=red, **STRICT))
        qp = h.assemble(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        out[red] = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
        h.close()
    a, b = out[1], out[0]
    assert np.array_equal(a.status, b.status) and np.all(a.iters[:, 0] == 1) and np.all(b.iters[:, 0] == 1)
    ok = a.status == 1
    keep = np.ones(5 * sc.N + 3, bool)
    keep[[3 * sc.N + 1, 5 * sc.N + 2]] = False              # e_psi_N and kappa_{N-1} are cost free
    assert np.max(np.abs(a.z[ok][:, keep] - b.z[ok][:, keep])) <= 1e-9
    prim, stat, comp = T.kkt_batch(qp[:, ok, :], sc.N, a.z[ok], a.y[ok])
    assert max(prim.max(), stat.max(), comp.max()) <= 1e-8
    inf = ~ok
    if inf.any():
        fk, _, _ = T.farkas_batch(qp[:, inf, :], sc.N, a.y[inf])
        assert fk.all()
    assert a.iters[ok, 1].mean() <= b.iters[ok, 1].mean() + 0.2


def test_deferred_tail_launch_gives_the_same_results(track):
    """The launcher stops enqueueing the (4.8 us, usually empty) tail launch of the reduced-native kernels once a launch it
    has seen the outcome of left no tail, and runs it late - at the next sync / download / upload - when a launch does
    leave one (mpmpc_hip.hip, observe_tail).  Whatever the history of the handle, the answers are those of a fresh one."""
    N, B = 30, 512
    feas = scenarios.make(2, track, B=B, N=N)             # every instance certified by the first kernel: no tail
    hard = scenarios.make(4, track, B=B, N=N)             # ~9 % infeasible: a tail of ~45 instances
    assert feas.weights == hard.weights

    def fresh(sc):
        h = _handle(track, N, sc.weights, B)
        return h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)

    ref_f, ref_h = fresh(feas), fresh(hard)
    assert (ref_h.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 10 and (ref_f.status == mpmpc.SOLVED).all()

    def same(a, b):
        assert np.array_equal(a.status, b.status) and np.array_equal(a.iters, b.iters)
        assert np.array_equal(a.z, b.z) and np.array_equal(a.u0, b.u0) and np.array_equal(a.resid, b.resid)

    h = _handle(track, N, feas.weights, B)
    h.set_outputs(True)
    # 1. learn "no tail" on the feasible batch, then meet the hard one with the tail launch deferred: resident path
    h.upload(feas.wp_id, feas.x0, feas.cc_prev, feas.lb, feas.ub)
    for _ in range(3):
        h.solve_resident(B)
        h.sync()
    same(h.download(B, want_y=True), ref_f)
    h.upload(hard.wp_id, hard.x0, hard.cc_prev, hard.lb, hard.ub)
    h.solve_resident(B)                                   # deferred: the handle expects no tail
    got = h.download(B, want_y=True)                      # ... and runs it here
    same(got, ref_h)
    assert np.array_equal(got.y, ref_h.y)
    # 2. now the handle launches the tail eagerly again; several launches in a row, one sync
    for _ in range(3):
        h.solve_resident(B)
    h.sync()
    same(h.download(B), ref_h)
    # 3. back to the feasible batch (eager, then deferred again), then the hard one through the host-buffer call
    h.upload(feas.wp_id, feas.x0, feas.cc_prev, feas.lb, feas.ub)
    for _ in range(2):
        h.solve_resident(B)
        h.sync()
    same(h.download(B), ref_f)
    same(h.solve(hard.wp_id, hard.x0, hard.cc_prev, hard.lb, hard.ub), ref_h)
    same(h.solve(feas.wp_id, feas.x0, feas.cc_prev, feas.lb, feas.ub), ref_f)
    # 4. a timed launch that turns out to need its deferred tail
    h.upload(feas.wp_id, feas.x0, feas.cc_prev, feas.lb, feas.ub)
    h.solve_resident(B)
    h.sync()
    h.upload(hard.wp_id, hard.x0, hard.cc_prev, hard.lb, hard.ub)
    ms = h.solve_resident_timed(B)
    assert ms[1] > 0
    same(h.download(B), ref_h)


@pytest.mark.parametrize("cfgid,B", [(2, 1024), (4, 2048), (3, 512)])
def test_pipelined_resident_launches_give_the_results_of_launches_done_one_at_a_time(cfgid, B, track):
    """mpmpc_solve_resident takes the launch slots of the handle in turn (stream, output block, tail lists; four by default,
    mpmpc_set_pipeline): launch k + 1 .. k + 3 run beside launch k.  Whatever the number of launches in flight and whatever
    came before on the handle, the results of the last launch are, bit for bit, those of the same launch done alone
    (mpmpc_solve: upload, launch, download) - through tails (config 4), deferred tails, re-uploads between launches, the
    host-buffer call, and for every pipeline depth; a handle held at ONE launch in flight packs its waves differently below
    1 024 instances (latency instead of throughput) and is compared with its own kind."""
    sc = scenarios.make(cfgid, track, B=B)
    other = scenarios.make(cfgid, track, B=B)
    perm = np.random.default_rng(cfgid).permutation(B)
    other.wp_id, other.x0, other.cc_prev, other.lb, other.ub = (a[perm] for a in (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))

    def same(a, b):
        assert np.array_equal(a.status, b.status) and np.array_equal(a.iters, b.iters)
        assert np.array_equal(a.z, b.z) and np.array_equal(a.u0, b.u0) and np.array_equal(a.resid, b.resid) and np.array_equal(a.y, b.y)

    one = _handle(track, sc.N, sc.weights, B)
    ref, ref_o = [one.solve(x.wp_id, x.x0, x.cc_prev, x.lb, x.ub, want_y=True) for x in (sc, other)]
    one.set_pipeline(1)
    ref1 = one.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    h = _handle(track, sc.N, sc.weights, B)
    h.set_outputs(True)
    for depth in (4, 2, 8, 3):
        h.set_pipeline(depth)
        for n_launch in (1, 2, 5, 9):
            h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
            for _ in range(n_launch):
                h.solve_resident(B)
            same(h.download(B, want_y=True), ref)                       # (waits for every slot)
            h.upload(other.wp_id, other.x0, other.cc_prev, other.lb, other.ub)      # the upload waits for what is in flight, too
            for _ in range(n_launch):
                h.solve_resident(B)
            h.sync()
            same(h.download(B, want_y=True), ref_o)
    # a number of launches that leaves the slots rotated: the host-buffer call and the staged call still see "the last launch"
    h.set_pipeline(4)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    h.solve_resident(B)
    same(h.solve(other.wp_id, other.x0, other.cc_prev, other.lb, other.ub, want_y=True), ref_o)
    h.solve_resident(B)                                              # (the batch of the host-buffer call is resident now)
    h.solve_resident(B)
    same(h.download(B, want_y=True), ref_o)
    # one launch at a time on request
    h.set_pipeline(1)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(3):
        h.solve_resident(B)
    same(h.download(B, want_y=True), ref1)
    ok = ref.status == 1
    assert np.array_equal(ref.status, ref1.status) and np.max(np.abs(ref.u0[ok] - ref1.u0[ok])) <= 1e-9      # (the two packings agree to rounding)
    for bad in (0, 9):
        with pytest.raises(mpmpc.MpmpcError):
            h.set_pipeline(bad)
    h.close()
    one.close()


def test_resident_launch_between_the_halves_of_a_staged_call(track):
    """a resident (or timed) launch issued between mpmpc_staged_begin and mpmpc_staged_end ends the begun call
    first - its deferred tail runs and its results are complete in the staging block - instead of overwriting its state."""
    N, B = 30, 300
    hard = scenarios.make(4, track, B=B, N=N)
    feas = scenarios.make(2, track, B=B, N=N)
    h = _handle(track, N, hard.weights, B, table="obstacles")
    ref_h = h.solve(hard.wp_id, hard.x0, hard.cc_prev, hard.lb, hard.ub, want_y=True)
    ref_f = h.solve(feas.wp_id, feas.x0, feas.cc_prev, feas.lb, feas.ub, want_y=True)
    assert (ref_h.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 5
    g = _handle(track, N, hard.weights, B, table="obstacles")
    v = g.staging(B)

    def fill(sc):
        v["wp_id"][:] = sc.wp_id; v["x0"][:] = sc.x0; v["cc_prev"][:] = sc.cc_prev; v["lb"][:] = sc.lb; v["ub"][:] = sc.ub

    fill(feas)
    for _ in range(2):                                   # the handle learns "no tail": the next tail launch is deferred
        g.solve_staged(B, want_z=True, want_y=True)
    for launch in ("resident", "timed", "outputs"):
        fill(hard)
        g.staged_begin(B, want_z=True, want_y=True)      # leaves a tail whose launch was not enqueued
        if launch == "resident":
            g.solve_resident(B)
        elif launch == "timed":
            g.solve_resident_timed(B)
        else:
            g.set_outputs(True)
        # the begun call has been ended by the other call: complete results, no MPMPC_UNSOLVED left behind
        assert np.array_equal(v["status"], ref_h.status) and np.array_equal(v["z"], ref_h.z) and np.array_equal(v["y"], ref_h.y)
        g.staged_end()                                   # (nothing left to do)
        got = g.download(B, want_y=True)
        assert np.array_equal(got.status, ref_h.status) and np.array_equal(got.z, ref_h.z)
        fill(feas)
        for _ in range(2):
            g.solve_staged(B, want_z=True, want_y=True)
        assert np.array_equal(v["status"], ref_f.status)
    g.close()
    h.close()


def test_staged_host_path_is_the_host_buffer_path_without_its_copies(track):
    """mpmpc_staging / mpmpc_solve_staged: the caller fills the handle's page-locked staging blocks and reads the results
    there.  Same answers as mpmpc_solve, bit for bit - with corridor rows and with the corridor table, with and without z / y,
    and through a deferred tail launch."""
    N, B = 30, 300
    hard = scenarios.make(4, track, B=B, N=N)
    feas = scenarios.make(2, track, B=B, N=N)
    h = _handle(track, N, hard.weights, B, table="obstacles")
    ref = {}
    for name, sc in (("hard", hard), ("feas", feas)):
        ref[name] = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    ref_tab = h.solve(hard.wp_id, hard.x0, hard.cc_prev, want_y=True)               # rows from the corridor table
    assert (ref["hard"].status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 5

    g = _handle(track, N, hard.weights, B, table="obstacles")
    v = g.staging(B)

    def fill(sc, rows=True):
        v["wp_id"][:] = sc.wp_id; v["x0"][:] = sc.x0; v["cc_prev"][:] = sc.cc_prev
        if rows:
            v["lb"][:] = sc.lb; v["ub"][:] = sc.ub

    def same(r, z=True, y=True):
        assert np.array_equal(v["status"], r.status) and np.array_equal(v["iters"], r.iters)
        assert np.array_equal(v["u0"], r.u0) and np.array_equal(v["resid"], r.resid)
        if z:
            assert np.array_equal(v["z"], r.z)
        if y:
            assert np.array_equal(v["y"], r.y)

    fill(hard)
    g.solve_staged(B, want_z=True, want_y=True)
    same(ref["hard"])
    fill(feas)
    for _ in range(2):                                   # the handle learns "no tail" ...
        g.solve_staged(B, want_z=True, want_y=False)
    same(ref["feas"], y=False)
    fill(hard)                                           # ... and meets a batch that leaves one
    g.solve_staged(B, want_z=False, want_y=False)
    same(ref["hard"], z=False, y=False)
    g.solve_staged(B, with_rows=False, want_z=True, want_y=True)
    same(ref_tab)
    # the two halves, two handles in flight: begin / begin / end / end
    g2 = _handle(track, N, hard.weights, B, table="obstacles")
    v2 = g2.staging(B)
    fill(feas)
    v2["wp_id"][:] = hard.wp_id; v2["x0"][:] = hard.x0; v2["cc_prev"][:] = hard.cc_prev; v2["lb"][:] = hard.lb; v2["ub"][:] = hard.ub
    g.staged_begin(B, want_z=True, want_y=False)
    g2.staged_begin(B, want_z=True, want_y=True)
    with pytest.raises(mpmpc.MpmpcError):
        g.staged_begin(B)                                # one at a time per handle
    g.staged_end()
    g2.staged_end()
    same(ref["feas"], y=False)
    assert np.array_equal(v2["status"], ref["hard"].status) and np.array_equal(v2["z"], ref["hard"].z) and np.array_equal(v2["y"], ref["hard"].y)
    g2.staged_begin(B, want_z=False)
    g2.sync()                                            # any other call ends a begun one
    assert np.array_equal(v2["u0"], ref["hard"].u0)
    v["wp_id"][0] = -1
    with pytest.raises(mpmpc.MpmpcError):
        g.solve_staged(B)


@pytest.mark.parametrize("native", [1, 0])
def test_empty_speed_box_is_reported_at_once_on_device(native, track):
    """umin[0] above the curvature-dependent speed cap (src/MPC.py:111-113): an empty interval row.  The instance is reported
    infeasible (zero ray, the gap as violation) by the reduced-native kernels and by the general ones alike, it spends no
    iteration on it (it used to sit through the whole ADMM run in the general kernels), and its neighbours in the
    batch are solved as if it were not there."""
    sc = scenarios.make(2, track, B=8)
    Q, R, QN = scenarios.WEIGHTS["stock"]
    umin = scenarios.UMIN.copy()
    umin[0] = 0.9
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, umin, scenarios.UMAX, scenarios.AY_MAX,
                            scenarios.CAR_LENGTH, max_batch=8)
    cc = sc.cc_prev.copy()
    cc[:4, 1::2], cc[:4, 0::2] = 0.6, 1.0            # large predicted steering: speed cap 0.12 < umin 0.9
    cc[4:] = 0.0
    h = mpmpc.Handle(cfg, mpmpc.default_settings(native=native))
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    sol = h.solve(sc.wp_id, sc.x0, cc, sc.lb, sc.ub, want_y=True)
    assert np.all(sol.status[:4] == mpmpc.PRIMAL_INFEASIBLE) and np.all(sol.y[:4] == 0.0) and np.all(sol.resid[:4, 0] > 0.5)
    assert np.all(sol.iters[:4, 1] == 0) and np.all(sol.iters[:4, 0] <= 1)
    assert np.all(sol.status[4:] == 1)
    alone = h.solve(sc.wp_id[4:], sc.x0[4:], cc[4:], sc.lb[4:], sc.ub[4:])
    assert np.array_equal(alone.status, sol.status[4:]) and np.max(np.abs(alone.u0 - sol.u0[4:])) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("cfgid,B,accept", [(4, 4096, 1), (5, 2048, 1), (4, 1024, 0)])
def test_reduced_native_tail_kernel_gives_the_general_kernels_answers(cfgid, B, accept, track):
    """K2p (mpmpc_reduced_tail_kernel) takes the tail of a batch launch before the general kernel does.  Its
    one-instance-per-wave form (mpmpc_set_tail_kernel(h, 2)) against the same launch with mpmpc_set_tail_kernel(h, 0) - the
    general kernel on the whole tail, the sequence of rounds 2 - 3: statuses identical, points / multipliers / residuals equal
    to rounding (same interior point, same scaling), for one launch at a time and for pipelined resident launches.  The default
    form (two instances per wave) against both: same statuses, certified optima to 1e-9, least-violation points to 1e-6."""
    sc = scenarios.make(cfgid, track, B=B)
    st = mpmpc.default_settings(phase1_accept=accept)
    res = {}
    for lean in (2, 0, 1):
        h = _handle(track, sc.N, sc.weights, B, settings=st)
        h.set_tail_kernel(lean)
        h.set_outputs(True)
        one = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
        h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
        for _ in range(7):
            h.solve_resident(B)
        many = h.download(B, want_y=True)
        assert np.array_equal(one.status, many.status)
        np.testing.assert_allclose(many.z, one.z, rtol=0, atol=1e-9)      # (a packed wave's partner differs between the two paths)
        res[lean] = one
    a, b, dflt = res[2], res[0], res[1]
    assert np.array_equal(dflt.status, b.status) and np.array_equal(dflt.iters[:, 0], b.iters[:, 0])
    okd = dflt.status == 1
    np.testing.assert_allclose(dflt.z[okd], b.z[okd], rtol=0, atol=1e-9)
    np.testing.assert_allclose(dflt.z, b.z, rtol=0, atol=1e-6)
    np.testing.assert_allclose(dflt.resid[:, 0], b.resid[:, 0], rtol=1e-6, atol=1e-9)
    assert np.array_equal(a.status, b.status)
    assert (a.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 20
    if accept:
        assert (a.status == mpmpc.SOLVED_INACCURATE).sum() >= 5
    np.testing.assert_allclose(a.z, b.z, rtol=0, atol=1e-12)
    np.testing.assert_allclose(a.u0, b.u0, rtol=0, atol=1e-12)
    scale = np.maximum(1.0, np.abs(b.y).max(axis=1, keepdims=True))
    assert np.max(np.abs(a.y - b.y) / scale) <= 1e-12
    np.testing.assert_allclose(a.resid, b.resid, rtol=1e-9, atol=1e-15)
    assert np.array_equal(a.iters[:, 0], b.iters[:, 0])


@pytest.mark.gpu
def test_two_tail_instances_per_wave_on_device(track, emu):
    """The default form of the tail kernel (two instances per wave, an odd tail included) against mpmpc_set_tail_kernel(h, 2),
    one instance per wave: the same statuses, points within 1e-6 (certified optima 1e-9), and what the emulation of the same
    kernel gives."""
    sc = scenarios.make(4, track, B=2049)
    res = {}
    for mode in (2, 1):
        h = _handle(track, sc.N, sc.weights, sc.B)
        h.set_tail_kernel(mode)
        res[mode] = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    a, b = res[2], res[1]
    ok = a.status == 1
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    e, _ = emu.solve_launch(cfg, mpmpc.default_settings(), qp, G=32)
    # (least-violation points of infeasible instances sit in flat directions of phase 1's problem: the device's reciprocal
    #  seeds and the emulation's divisions end micrometres apart there; certified optima agree to 1e-9)
    assert np.array_equal(e.status, b.status) and np.max(np.abs(e.z[ok] - b.z[ok])) <= 1e-9 and np.max(np.abs(e.z - b.z)) <= 1e-4
    assert np.array_equal(a.status, b.status) and (a.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 100 and (a.status == 2).sum() >= 5
    np.testing.assert_allclose(b.z[ok], a.z[ok], rtol=0, atol=1e-9)
    np.testing.assert_allclose(b.z, a.z, rtol=0, atol=1e-6)
    np.testing.assert_allclose(b.resid[:, 0], a.resid[:, 0], rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_what_the_tail_kernel_leaves_reaches_the_general_kernel_also_when_deferred(track):
    """Second level of the deferred tail: the general kernel's launch on what K2p leaves is not enqueued while the launches
    the host has seen leave nothing there.  With as_rounds = 0 nothing can be certified by an attempt, so every feasible
    instance falls through K2p to the general kernel's full OSQP run: a handle that has learned "K2p leaves nothing" on the
    default settings must still deliver those results (late, at the download), exactly as a fresh handle does."""
    B = 192
    sc = scenarios.make(4, track, B=B)
    hard = mpmpc.default_settings(as_rounds=0)
    fresh = _handle(track, sc.N, sc.weights, B, settings=hard)
    ref = fresh.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    assert (ref.iters[:, 0] > 25).sum() >= 100 and (ref.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 3
    old = _handle(track, sc.N, sc.weights, B, settings=hard)
    old.set_tail_kernel(False)
    gen = old.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    assert np.array_equal(gen.status, ref.status) and np.array_equal(gen.iters[:, 0], ref.iters[:, 0])
    np.testing.assert_allclose(ref.z, gen.z, rtol=0, atol=1e-6)          # (least-violation points: the default form's phase 1)

    def same(a, b):
        assert np.array_equal(a.status, b.status) and np.array_equal(a.iters, b.iters)
        assert np.array_equal(a.z, b.z) and np.array_equal(a.u0, b.u0) and np.array_equal(a.resid, b.resid) and np.array_equal(a.y, b.y)

    h = _handle(track, sc.N, sc.weights, B)
    h.set_outputs(True)
    dflt = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    assert (dflt.iters[:, 0] == 1).all()
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(3):                                   # the tail is there every time, K2p leaves nothing: learned
        h.solve_resident(B)
        h.sync()
    same(h.download(B, want_y=True), dflt)
    h.set_settings(hard)
    h.solve_resident(B)                                  # K2p's leftovers: their launch is deferred ...
    same(h.download(B, want_y=True), ref)                # ... and runs here
    for _ in range(3):                                   # eager again, pipelined
        h.solve_resident(B)
    h.sync()
    same(h.download(B, want_y=True), ref)
    h.set_settings(mpmpc.default_settings())
    for _ in range(2):
        h.solve_resident(B)
        h.sync()
    same(h.download(B, want_y=True), dflt)
    same(h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True), dflt)


@pytest.mark.gpu
@pytest.mark.parametrize("N,B", [(40, 600), (50, 300), (32, 257)])
def test_reduced_native_tail_kernel_at_horizons_above_31(N, B, track, emu):
    """mpmpc_reduced_tail_kernel<64,32>: the tail of the stock-type weights at horizons 32 .. 63 (one lane per stage).  Against
    the general kernel on the whole tail and against the emulation of the same kernel."""
    sc = scenarios.make(4, track, B=B, N=N)
    res = {}
    for mode in (1, 0):
        h = _handle(track, sc.N, sc.weights, B)
        h.set_tail_kernel(mode)
        res[mode] = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, want_y=True)
    a, b = res[1], res[0]
    assert np.array_equal(a.status, b.status) and np.array_equal(a.iters[:, 0], b.iters[:, 0])
    assert (a.status == mpmpc.PRIMAL_INFEASIBLE).sum() >= 3 and (a.iters[:, 0] == 1).all()
    ok = a.status == 1
    np.testing.assert_allclose(a.z[ok], b.z[ok], rtol=0, atol=1e-9)
    np.testing.assert_allclose(a.z, b.z, rtol=0, atol=1e-5)
    cfg = T.stock_config(sc.N, sc.weights)
    qp = emu.assemble(cfg, track, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    e, _ = emu.solve_launch(cfg, mpmpc.default_settings(), qp, G=64)
    assert np.array_equal(e.status, a.status) and np.max(np.abs(e.z[ok] - a.z[ok])) <= 1e-9
    inf = a.status == mpmpc.PRIMAL_INFEASIBLE
    assert T.farkas_batch(qp[:, inf, :], sc.N, a.y[inf])[0].all()
