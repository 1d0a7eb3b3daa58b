"""MPC.get_control() host logic (src/MPC.py:161-222) against the reference's own closed loop
(golden G6: src/simulation.py's while-loop driven through the reference classes, teacher-forced
per step).  The solver behind the controller is the CPU emulation of the HIP kernels here; the
-m gpu twin runs the same lap through libmpmpc.so."""
import numpy as np
import pytest

import mpc_np as M
import mpmpc
import mpmpc_testlib as T
import scenarios
from map import Map
from MPC import MPC
from reference_path import ReferencePath
from spatial_bicycle_models import BicycleModel, TemporalState

G = M.GOLDEN


def build_world(obstacles=True):
    g1 = np.load(G + "/g1_path_sim_track.npz")
    g2 = np.load(G + "/g2_speed_profile.npz")
    h, w = g1["grid_shape"]
    grid = np.unpackbits(g1["grid_obstacles" if obstacles else "grid_free"])[:h * w].reshape(h, w).astype(np.int8)
    m = Map.from_grid(grid, origin=[-1, -2], resolution=0.005)
    rp = ReferencePath.from_tables(m, g1["x"], g1["y"], g1["psi"], g1["kappa"], circular=True, v_ref=g2["v_ref"],
                                   border_ub=g1["border_ub"], border_lb=g1["border_lb"],
                                   ub_static=g1["ub_static"], lb_static=g1["lb_static"])
    car = BicycleModel(reference_path=rp, length=0.12, width=0.06, Ts=0.05)
    return m, rp, car


def make_mpc(car, N, backend=None, corridor="host"):
    from scipy import sparse
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    sc = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    if backend == "emu":
        cfg = T.stock_config(N)
        backend = T.EmuBackend(cfg, mpmpc.default_settings())
    return MPC(car, N, Q, R, QN, sc, ic, 4.0, backend=backend, corridor=corridor)


@pytest.mark.parametrize("N,stride", [(10, 1), (30, 3)])
def test_get_control_reproduces_reference_lap(N, stride):
    g = np.load(G + "/g6_closed_loop_N%d.npz" % N)
    m, rp, car = build_world()
    mpc = make_mpc(car, N, "emu")
    steps = range(0, g["s"].size, stride)
    n_inf = 0
    for t in steps:
        car.s = float(g["s"][t])
        car.temporal_state = TemporalState(*g["pose"][t])
        mpc.current_control = g["cc_prev"][t].copy()
        mpc.infeasibility_counter = int(g["counter"][t - 1]) if t > 0 else 0
        u = mpc.get_control()
        assert car.wp_id == g["wp_id"][t]
        assert np.allclose(np.array(car.spatial_state[:]), g["x0"][t], atol=1e-13)
        assert mpc.infeasibility_counter == g["counter"][t]
        ref_ok = g["status"][t] > 0
        assert (mpc.last_status > 0) == ref_ok, (t, mpc.last_status, g["status"][t])
        assert np.max(np.abs(u - g["u"][t])) <= 1e-6, (t, u, g["u"][t])
        if ref_ok:
            d = np.abs(mpc.current_control - g["cc_next"][t])
            d[-1] = 0.0                                   # kappa_{N-1} is cost free
            assert d.max() <= 1e-6
            # MPC.update_prediction (src/MPC.py:224-248): the reference's own (x_pred, y_pred) of this step
            px, py = mpc.current_prediction
            assert len(px) == len(py) == N - 2
            assert np.max(np.abs(np.array(px) - g["pred_x"][t])) <= 1e-6 and np.max(np.abs(np.array(py) - g["pred_y"][t])) <= 1e-6
        else:
            n_inf += 1
            assert np.array_equal(mpc.current_control, g["cc_prev"][t])
    assert n_inf > 0           # the fallback branch (src/MPC.py:208-216) was exercised


def test_exit_after_n_minus_one_infeasible_steps():
    """src/MPC.py:218-220: the reference's lap at N=10 ends with exit(1); so does ours."""
    g = np.load(G + "/g6_closed_loop_N10.npz")
    assert bool(g["exited"][0])
    m, rp, car = build_world()
    mpc = make_mpc(car, 10, "emu")
    t = g["s"].size - 1
    # replay the recorded last good state, then drive freely: the run must stop with SystemExit
    car.s = float(g["s"][t])
    car.temporal_state = TemporalState(*g["pose"][t])
    mpc.current_control = g["cc_prev"][t].copy()
    mpc.infeasibility_counter = int(g["counter"][t - 1])
    with pytest.raises(SystemExit):
        for _ in range(12):
            u = mpc.get_control()
            car.drive(u)


def test_rejects_unsupported_weights_and_missing_profile():
    m, rp, car = build_world()
    from scipy import sparse
    ic = {'umin': np.array([0.0, -1.0]), 'umax': np.array([1.0, 1.0])}
    sc = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    Qbad = np.array([[1.0, 0.1, 0], [0.1, 0, 0], [0, 0, 0]])
    with pytest.raises(ValueError):
        MPC(car, 10, Qbad, sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0, 0]), sc, ic, 4.0, backend=object())
    with pytest.raises(ValueError):
        MPC(car, 2, sparse.diags([1.0, 0, 0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0, 0]), sc, ic, 4.0,
            backend=object())


def test_product_fails_loudly_without_device():
    """No CPU fallback behind the product classes: constructing the controller without a HIP
    device (this container) must raise, not silently compute elsewhere."""
    if mpmpc.device_count() > 0:
        pytest.skip("a GPU is present")
    m, rp, car = build_world()
    with pytest.raises((mpmpc.MpmpcError, RuntimeError)):
        make_mpc(car, 10, backend=None)
