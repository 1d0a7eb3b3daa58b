"""Closed-loop rollout: the per-car code of the K3 kernels on the host against the
reference's own closed-loop trace (golden G6), and - on the GPU - B cars rolled out on the device
against the host loop of src/simulation.py:134-140 run with our classes."""
import ctypes as C

import numpy as np
import pytest

import mpc_np as M
import mpmpc
import mpmpc_testlib as T

dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


def test_localise_and_advance_reproduce_reference_trace(emu):
    g = np.load(M.GOLDEN + "/g6_closed_loop_N30.npz")
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    N, L, Ts = 30, 0.12, 0.05
    cum = np.ascontiguousarray(np.cumsum(g1["segment_lengths"]))
    gx, gy, gpsi = (np.ascontiguousarray(g1[k]) for k in ("x", "y", "psi"))
    steps = g["s"].size
    for t in range(steps - 1):
        pose = np.ascontiguousarray(g["pose"][t])
        x0 = np.zeros(3)
        wp = emu.lib.emu_localise(C.c_int(200), _d(cum), _d(gx), _d(gy), _d(gpsi), C.c_double(float(g["s"][t])), _d(pose), _d(x0))
        assert wp == g["wp_id"][t] and np.allclose(x0, g["x0"][t], rtol=0, atol=1e-14)
        # feed the reference's own solution of this step and compare the plant / plan update
        cc = np.ascontiguousarray(g["cc_prev"][t].copy())
        z = np.ascontiguousarray(np.nan_to_num(g["z"][t]))
        counter = C.c_int(int(g["counter"][t - 1]) if t > 0 else 0)
        s = C.c_double(float(g["s"][t]))
        u = np.zeros(2)
        alive = emu.lib.emu_advance(C.c_int(N), C.c_double(L), C.c_double(Ts), C.c_int(int(g["status"][t])), _d(z), _d(cc),
                                    C.byref(counter), _d(x0), C.c_double(float(g1["kappa"][wp])), _d(pose), C.byref(s), _d(u))
        assert alive == 1 and counter.value == g["counter"][t]
        assert np.allclose(u, g["u"][t], rtol=0, atol=1e-15)
        assert np.allclose(cc, g["cc_next"][t], rtol=0, atol=1e-15)
        assert abs(s.value - g["s"][t + 1]) <= 1e-14 and np.allclose(pose, g["pose"][t + 1], rtol=0, atol=1e-14)
    # past the end of the path: reported, not wrapped
    assert emu.lib.emu_localise(C.c_int(200), _d(cum), _d(gx), _d(gy), _d(gpsi), C.c_double(float(cum[-1]) + 1.0),
                                _d(pose), _d(x0)) == -1


def test_advance_ends_the_run_after_n_minus_one_fallbacks(emu):
    N = 6
    cc = np.ascontiguousarray(np.arange(12, dtype=float) / 10)
    counter = C.c_int(0)
    pose, x0, u, z = np.zeros(3), np.zeros(3), np.zeros(2), np.zeros(5 * N + 3)
    s = C.c_double(0.0)
    alive = []
    for _ in range(N - 1):
        alive.append(emu.lib.emu_advance(C.c_int(N), C.c_double(0.12), C.c_double(0.05), C.c_int(-3), _d(z), _d(cc),
                                         C.byref(counter), _d(x0), C.c_double(0.0), _d(pose), C.byref(s), _d(u)))
    assert alive == [1, 1, 1, 1, 0] and counter.value == N - 1       # src/MPC.py:218-220
    assert u[0] == cc[2 * (N - 1)] and u[1] == cc[2 * (N - 1) + 1]


@pytest.mark.gpu
def test_device_rollout_matches_host_loop():
    import test_host_mpc as H
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    g3 = np.load(M.GOLDEN + "/g3_corridor.npz")
    N, steps = 30, 25
    starts = np.array([0, 12, 37, 61, 88, 120, 150, 171])
    cum = np.cumsum(g1["segment_lengths"])
    B = starts.size
    # ---- host: one controller per car, the reference's loop with our classes
    host = []
    for w in starts:
        m, rp, car = H.build_world()
        mpc = H.make_mpc(car, N)
        car.s = float(cum[w])
        car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi = rp.waypoints[w].x, rp.waypoints[w].y, rp.waypoints[w].psi
        for _ in range(steps):
            u = mpc.get_control()
            car.drive(u)
        host.append((car.s, car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi, u[0], u[1]))
    host = np.array(host)
    # ---- device: all cars at once
    cfg = T.stock_config(N, max_batch=B)
    h = mpmpc.Handle(cfg)
    tr = __import__("scenarios").sim_track()
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_corridor(g3["ub_obstacles"], g3["lb_obstacles"])
    h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
    poses = np.stack([g1["x"][starts], g1["y"][starts], g1["psi"][starts]], axis=1)
    h.rollout_init(0.05, cum, cum[starts], poses)
    h.rollout_step(steps)
    st = h.rollout_state()
    assert np.all(st["alive"] == 1)
    assert np.max(np.abs(st["s"] - host[:, 0])) <= 1e-8
    assert np.max(np.abs(st["pose"] - host[:, 1:4])) <= 1e-8
    assert np.max(np.abs(st["u"] - host[:, 4:6])) <= 1e-6
    h.close()


@pytest.mark.gpu
def test_device_rollout_matches_host_loop_at_a_long_horizon(emu):
    """The closed loop at N = 70 - one instance per WORKGROUP of two wavefronts on the device (reduced-native solver, general
    kernel on its tail), cold starts: every car follows the reference's own loop run with our host classes."""
    import test_host_mpc as H
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    tr = __import__("scenarios").sim_track()
    N, steps = 70, 6
    tw = T.wide_track(tr, emu, N)                   # corridor tables with N columns (golden G3 has 50; bit-equal on those)
    starts = np.array([5, 61, 120, 171])
    cum = np.cumsum(g1["segment_lengths"])
    B = starts.size
    host = []
    for w in starts:
        m, rp, car = H.build_world()
        mpc = H.make_mpc(car, N)
        car.s = float(cum[w])
        car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi = rp.waypoints[w].x, rp.waypoints[w].y, rp.waypoints[w].psi
        for _ in range(steps):
            u = mpc.get_control()
            car.drive(u)
        host.append((car.s, car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi, u[0], u[1]))
    host = np.array(host)
    cfg = T.stock_config(N, max_batch=B)
    h = mpmpc.Handle(cfg)
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_corridor(tw.ub_obstacles, tw.lb_obstacles)
    h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
    poses = np.stack([g1["x"][starts], g1["y"][starts], g1["psi"][starts]], axis=1)
    h.rollout_init(0.05, cum, cum[starts], poses)
    h.rollout_step(steps)
    st = h.rollout_state()
    h.close()
    assert np.all(st["alive"] == 1)
    assert np.max(np.abs(st["s"] - host[:, 0])) <= 1e-8
    assert np.max(np.abs(st["pose"] - host[:, 1:4])) <= 1e-8
    assert np.max(np.abs(st["u"] - host[:, 4:6])) <= 1e-6


@pytest.mark.gpu
def test_rollout_warm_start_changes_nothing_but_the_time():
    """The closed loop with the warm start (previous step's shifted active set first, default) and without it:
    same trajectories, same statuses; most steps of the warm run need no ADMM / interior-point iteration."""
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    g3 = np.load(M.GOLDEN + "/g3_corridor.npz")
    N, steps, B = 30, 40, 96
    tr = __import__("scenarios").sim_track()
    cum = np.cumsum(g1["segment_lengths"])
    starts = np.random.default_rng(3).integers(0, 200, B)
    poses = np.stack([g1["x"][starts], g1["y"][starts], g1["psi"][starts]], axis=1)
    out = {}
    for warm in (False, True):
        h = mpmpc.Handle(T.stock_config(N, max_batch=B))
        h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        h.set_corridor(g3["ub_free"], g3["lb_free"])
        h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
        h.rollout_warm_start(warm)
        h.rollout_init(0.05, cum, cum[starts], poses)
        h.rollout_step(steps)
        out[warm] = (h.rollout_state(), h.download(B))
        h.close()
    (a, sa), (b, sb) = out[False], out[True]
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["alive"], b["alive"])
    assert np.max(np.abs(a["s"] - b["s"])) <= 1e-8 and np.max(np.abs(a["pose"] - b["pose"])) <= 1e-8
    assert np.all(sa.iters[:, 0] >= 1)                    # cold: every step runs the early attempt
    assert np.mean(sb.iters[:, 0] == 0) > 0.7             # warm: most cars certified straight from the guess


@pytest.mark.gpu
def test_rollout_warm_start_default_is_used_where_it_pays():
    """Default ("auto"): fleets of at most 16 cars and packed launches (more than 1024 cars) start from the previous
    step's active sets, fleets in between do not (a step ends with its slowest car there)."""
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    g3 = np.load(M.GOLDEN + "/g3_corridor.npz")
    tr = __import__("scenarios").sim_track()
    cum = np.cumsum(g1["segment_lengths"])
    for B, expect_warm in ((8, True), (96, False), (1100, True)):
        starts = np.random.default_rng(5).integers(0, 200, B)
        poses = np.stack([g1["x"][starts], g1["y"][starts], g1["psi"][starts]], axis=1)
        h = mpmpc.Handle(T.stock_config(30, max_batch=B))
        h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        h.set_corridor(g3["ub_free"], g3["lb_free"])
        h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
        h.rollout_init(0.05, cum, cum[starts], poses)          # no rollout_warm_start call: the default
        h.rollout_step(6)
        sol = h.download(B)
        h.close()
        hits = float(np.mean(sol.iters[:, 0] == 0))
        assert (hits > 0.5) if expect_warm else (hits == 0.0), (B, hits)


def _g6_handle(N, B, settings=None):
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    g3 = np.load(M.GOLDEN + "/g3_corridor.npz")
    tr = __import__("scenarios").sim_track()
    # (golden G6 was recorded with a solver that reports every proven infeasibility: phase1_accept = 0)
    h = mpmpc.Handle(T.stock_config(N, max_batch=B), settings or mpmpc.default_settings(phase1_accept=0))
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    h.set_corridor(g3["ub_obstacles"], g3["lb_obstacles"])
    h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
    return h, np.cumsum(g1["segment_lengths"])


@pytest.mark.gpu
@pytest.mark.parametrize("N", [10, 30])
def test_device_rollout_replays_the_reference_trace(N):
    """K3 on the GPU against the reference's OWN closed loop (golden G6: src/simulation.py:134-140 driven through the
    reference classes), teacher-forced: every recorded step becomes one car (its s, pose, previous plan and
    infeasibility counter), ONE rollout step runs localise -> solve -> advance for all of them, and each car must land
    where the reference's step landed: waypoint, spatial state, verdict, control, new plan, counter, next s and pose."""
    g = np.load(M.GOLDEN + "/g6_closed_loop_N%d.npz" % N)
    T_ = g["s"].size
    h, cum = _g6_handle(N, T_)
    prev_counter = np.concatenate([[0], g["counter"][:-1]]).astype(np.int32)
    h.rollout_warm_start(False)
    h.rollout_init(0.05, cum, g["s"], g["pose"], cc0=g["cc_prev"])
    h.rollout_set_counters(prev_counter)
    h.rollout_step(1)
    st = h.rollout_state()
    h.close()
    assert np.array_equal(st["wp_id"], g["wp_id"])
    assert np.max(np.abs(st["x0"] - g["x0"])) <= 1e-13
    ok = g["status"] > 0
    assert np.array_equal(st["status"] > 0, ok) and (~ok).sum() >= 5            # the fallback branch is exercised
    assert np.array_equal(st["counter"], g["counter"])
    assert np.max(np.abs(st["u"] - g["u"])) <= 1e-6
    d = np.abs(st["cc"] - g["cc_next"])
    d[:, -1] = 0.0                                                              # kappa_{N-1} is cost free
    assert d.max() <= 1e-6
    assert np.all(st["alive"] == 1)
    assert np.max(np.abs(st["s"][:-1] - g["s"][1:])) <= 1e-7
    assert np.max(np.abs(st["pose"][:-1] - g["pose"][1:])) <= 1e-7


@pytest.mark.gpu
def test_device_rollout_ends_like_the_reference_at_horizon_10():
    """Config 1's own horizon: the reference's lap at N = 10 ends with exit(1) after N - 1 consecutive infeasible
    steps (src/MPC.py:218-220).  A car put on the last recorded state ends the same way (alive = -1) on the device."""
    g = np.load(M.GOLDEN + "/g6_closed_loop_N10.npz")
    assert bool(g["exited"][0])
    t = g["s"].size - 1
    h, cum = _g6_handle(10, 1)
    h.rollout_init(0.05, cum, g["s"][t:t + 1], g["pose"][t:t + 1], cc0=g["cc_prev"][t:t + 1])
    h.rollout_set_counters(g["counter"][t - 1:t])
    h.rollout_step(12)
    st = h.rollout_state()
    h.close()
    assert st["alive"][0] == -1 and st["counter"][0] == 9


@pytest.mark.gpu
def test_rollout_state_is_guarded_against_interleaved_solves():
    """a single mpmpc_solve between rollout steps used to continue the rollout on garbage."""
    g = np.load(M.GOLDEN + "/g6_closed_loop_N30.npz")
    h, cum = _g6_handle(30, 4)
    h.rollout_init(0.05, cum, g["s"][:4], g["pose"][:4])
    h.rollout_step(2)
    ref = h.rollout_state()
    h.download(4)                                   # reading results is fine
    h.rollout_step(1)
    h.solve(g["wp_id"][:2].astype(np.int32), g["x0"][:2], g["cc_prev"][:2], g["lb"][:2], g["ub"][:2])
    with pytest.raises(mpmpc.MpmpcError, match="rollout_init"):
        h.rollout_step(1)
    with pytest.raises(mpmpc.MpmpcError):
        h.rollout_state()
    h.rollout_init(0.05, cum, ref["s"], ref["pose"], cc0=ref["cc"])
    h.rollout_step(1)
    assert np.all(h.rollout_state()["alive"] == 1)
    h.close()
