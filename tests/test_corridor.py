"""Corridor generation: the per-thread code of the K0 kernels, run on the host here
and on the device in the -m gpu test, against golden G3 - the tables the reference's own
update_path_constraints produced for every start waypoint of Sim_Track, with and without the
nine obstacles of src/simulation.py:40-48."""
import ctypes as C

import numpy as np
import pytest

import mpc_np as M
import mpmpc
import mpmpc_testlib as T
import scenarios

dp = C.POINTER(C.c_double)


def _golden():
    g1 = np.load(M.GOLDEN + "/g1_path_sim_track.npz")
    g3 = np.load(M.GOLDEN + "/g3_corridor.npz")
    h, w = g1["grid_shape"]
    grids = {k: np.ascontiguousarray(np.unpackbits(g1["grid_" + k])[:h * w].reshape(h, w).astype(np.int8))
             for k in ("free", "obstacles")}
    return g1, g3, grids


@pytest.mark.parametrize("key", ["free", "obstacles"])
def test_corridor_code_matches_reference_tables_bit_exact(key, emu):
    g1, g3, grids = _golden()
    grid = grids[key]
    sm = float(g3["safety_margin"][0])
    n, nc = 200, 50
    ub, lb, nseg = np.zeros((n, nc)), np.zeros((n, nc)), np.zeros(n, np.int32)
    arrs = [np.ascontiguousarray(g1[k], float) for k in ("x", "y", "psi", "ds_next")]
    bu, bl = np.ascontiguousarray(g1["border_ub"], float), np.ascontiguousarray(g1["border_lb"], float)
    bad = emu.lib.emu_corridor(C.c_int(grid.shape[0]), C.c_int(grid.shape[1]), grid.ctypes.data_as(C.POINTER(C.c_int8)),
                               C.c_double(-1.0), C.c_double(-2.0), C.c_double(0.005), C.c_int(n),
                               *[a.ctypes.data_as(dp) for a in arrs], C.c_int(1), bu.ctypes.data_as(dp),
                               bl.ctypes.data_as(dp), C.c_int(nc), C.c_double(2 * sm), C.c_double(sm),
                               ub.ctypes.data_as(dp), lb.ctypes.data_as(dp), nseg.ctypes.data_as(C.POINTER(C.c_int32)))
    assert bad == 0
    assert np.array_equal(ub, g3["ub_" + key]) and np.array_equal(lb, g3["lb_" + key])     # bit exact
    assert nseg.max() == (2 if key == "obstacles" else 1) and nseg.min() >= 1
    # a fully blocked first waypoint is reported, not silently bridged (the reference raises there)
    blocked = grid.copy()
    blocked[:] = 0
    bad = emu.lib.emu_corridor(C.c_int(grid.shape[0]), C.c_int(grid.shape[1]), blocked.ctypes.data_as(C.POINTER(C.c_int8)),
                               C.c_double(-1.0), C.c_double(-2.0), C.c_double(0.005), C.c_int(n),
                               *[a.ctypes.data_as(dp) for a in arrs], C.c_int(1), bu.ctypes.data_as(dp),
                               bl.ctypes.data_as(dp), C.c_int(nc), C.c_double(2 * sm), C.c_double(sm),
                               ub.ctypes.data_as(dp), lb.ctypes.data_as(dp), None)
    assert bad == n and np.all(np.isnan(ub))


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["free", "obstacles"])
def test_device_corridor_matches_reference_tables(key, track):
    g1, g3, grids = _golden()
    cfg = T.stock_config(30, max_batch=64)
    h = mpmpc.Handle(cfg)
    h.set_path(track.kappa, track.v_ref, track.ds_next)
    h.set_map(grids[key], (-1.0, -2.0), 0.005)
    h.set_path_geometry(g1["x"], g1["y"], g1["psi"], g1["border_ub"], g1["border_lb"])
    sm = float(g3["safety_margin"][0])
    ub, lb, bad = h.build_corridor(50, 2 * sm, sm)
    assert bad == 0
    # integer work (rasterisation, segment scan) is exact; the floats go through device sin/cos/atan2
    assert np.max(np.abs(ub - g3["ub_" + key])) <= 1e-13 and np.max(np.abs(lb - g3["lb_" + key])) <= 1e-13
    assert np.mean(ub == g3["ub_" + key]) > 0.9
    # the solve then reads the device-built table: same answers as with the rows passed in
    sc = scenarios.make(4 if key == "obstacles" else 2, track, B=64)
    a = h.solve(sc.wp_id, sc.x0, sc.cc_prev)                       # table built on the device
    b = h.solve(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)         # rows from the reference's tables
    assert np.array_equal(a.status, b.status)
    ok = a.status == 1
    assert np.max(np.abs(a.u0[ok] - b.u0[ok])) <= 1e-9
    h.close()


@pytest.mark.gpu
def test_batch_mpc_rebuilds_corridor_when_the_map_changes():
    """Dynamic map: stamp a new obstacle, rebuild on the device, compare with the host walk."""
    import test_host_mpc as H
    from map import Obstacle
    from MPC import BatchMPC
    from scipy import sparse
    m, rp, car = H.build_world(obstacles=False)
    Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]), 'umax': np.array([1.0, np.tan(0.66) / car.length])}
    sc = {'xmin': np.array([-np.inf] * 3), 'xmax': np.array([np.inf] * 3)}
    bm = BatchMPC(car, 30, Q, R, QN, sc, ic, 4.0, max_batch=16, corridor="device")
    sm = car.safety_margin
    for step in range(2):
        ub_h, lb_h = rp.corridor_table(30, 2 * sm, sm)
        ub_d, lb_d, bad = bm.handle.build_corridor(30, 2 * sm, sm)
        assert bad == 0 and np.max(np.abs(ub_d - ub_h)) <= 1e-13 and np.max(np.abs(lb_d - lb_h)) <= 1e-13
        m.add_obstacles([Obstacle(cx=-0.3, cy=-1.0, radius=0.08)])       # the map changes ...
        assert bm.update_corridor_from_map() == 0                        # ... and the table follows
    assert np.any(ub_d != np.load(M.GOLDEN + "/g3_corridor.npz")["ub_free"][:, :30])
    wp = np.arange(0, 160, 10).astype(np.int32)
    x0 = np.zeros((16, 3))
    u, plan, status, sol = bm.get_control_batch(wp, x0, np.zeros((16, 60)))
    assert np.all(np.isin(status, (1, 2, -3)))
