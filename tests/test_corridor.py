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


def _emu_corridor(emu, grid, origin, res, x, y, psi, ds, bu, bl, nc, min_width, sm):
    n = x.size
    ub, lb, nseg = np.zeros((n, nc)), np.zeros((n, nc)), np.zeros(n, np.int32)
    arrs = [np.ascontiguousarray(a, float) for a in (x, y, psi, ds)]
    bu, bl = np.ascontiguousarray(bu, float), np.ascontiguousarray(bl, float)
    rc = emu.lib.emu_corridor(C.c_int(grid.shape[0]), C.c_int(grid.shape[1]), grid.ctypes.data_as(C.POINTER(C.c_int8)),
                              C.c_double(origin[0]), C.c_double(origin[1]), C.c_double(res), C.c_int(n),
                              *[a.ctypes.data_as(dp) for a in arrs], C.c_int(1), bu.ctypes.data_as(dp),
                              bl.ctypes.data_as(dp), C.c_int(nc), C.c_double(min_width), C.c_double(sm),
                              ub.ctypes.data_as(dp), lb.ctypes.data_as(dp), nseg.ctypes.data_as(C.POINTER(C.c_int32)))
    return rc, ub, lb, nseg


def _edge_world(striped=False):
    """A 40 x 60 map whose straight path runs along its lower edge: the border lines end on row 0, so their
    anti-aliased neighbours step to row -1 (unchecked reads outside the grid)."""
    grid = np.ones((40, 60), np.int8)
    if striped:                                  # many thin walls across the border lines: more than 8 free segments
        grid[2:40:3, :] = 0
    n = 12
    x = 0.15 + 0.02 * np.arange(n)
    y = np.full(n, 0.1)
    psi = np.full(n, 0.3)                        # a tilted line: the rasterisation has off-axis neighbours
    ds = np.full(n, 0.02)
    bu = np.stack([x - 0.1, np.full(n, 0.39)], axis=1)       # upper border near the top rows
    bl = np.stack([x + 0.05, np.full(n, 0.004)], axis=1)     # lower border on row 0 (y in [0, 0.01))
    return grid, (0.0, 0.0), 0.01, x, y, psi, ds, bu, bl


def test_border_line_along_the_map_edge_reads_nothing_outside(emu):
    grid, origin, res, x, y, psi, ds, bu, bl = _edge_world()
    rc, ub, lb, nseg = _emu_corridor(emu, grid, origin, res, x, y, psi, ds, bu, bl, 5, 0.02, 0.01)
    assert rc == 0 and np.all(nseg == 1) and np.all(np.isfinite(ub)) and np.all(ub > lb)
    # a border point outside the map is an error, not a wrapped-around or out-of-bounds read
    bad = bl.copy()
    bad[3, 1] = -0.02
    rc, *_ = _emu_corridor(emu, grid, origin, res, x, y, psi, ds, bu, bad, 5, 0.02, 0.01)
    assert rc == -3000
    bad = bu.copy()
    bad[0, 0] = 0.75
    rc, *_ = _emu_corridor(emu, grid, origin, res, x, y, psi, ds, bad, bl, 5, 0.02, 0.01)
    assert rc == -3000


def test_more_free_segments_than_the_tables_hold_is_an_error(emu):
    grid, origin, res, x, y, psi, ds, bu, bl = _edge_world(striped=True)
    rc, *_ = _emu_corridor(emu, grid, origin, res, x, y, psi, ds, bu, bl, 5, 0.005, 0.001)
    assert rc == -3001                           # 13 free runs per line, COR_MAXSEG = 8: reported, not truncated


@pytest.mark.gpu
def test_device_corridor_edge_cases_and_errors(track):
    """The same three situations through libmpmpc.so: identical table on the map edge, MPMPC_E_ARG for a border point
    outside the map and for a line with more free segments than COR_MAXSEG."""
    grid, origin, res, x, y, psi, ds, bu, bl = _edge_world()
    emu = T.Emul()
    rc, ub_e, lb_e, _ = _emu_corridor(emu, grid, origin, res, x, y, psi, ds, bu, bl, 5, 0.02, 0.01)
    assert rc == 0
    cfg = T.stock_config(5, max_batch=4)
    h = mpmpc.Handle(cfg)
    kappa = np.zeros(x.size)
    h.set_path(kappa, np.ones(x.size), ds)
    h.set_map(grid, origin, res)
    h.set_path_geometry(x, y, psi, bu, bl)
    ub, lb, bad = h.build_corridor(5, 0.02, 0.01)
    assert bad == 0 and np.array_equal(ub, ub_e) and np.array_equal(lb, lb_e)
    bl2 = bl.copy()
    bl2[3, 1] = -0.02
    h.set_path_geometry(x, y, psi, bu, bl2)
    with pytest.raises(mpmpc.MpmpcError, match="outside the map"):
        h.build_corridor(5, 0.02, 0.01)
    grid2 = _edge_world(striped=True)[0]
    h.set_map(grid2, origin, res)
    h.set_path_geometry(x, y, psi, bu, bl)
    with pytest.raises(mpmpc.MpmpcError, match="free segments"):
        h.build_corridor(5, 0.005, 0.001)
    # a smaller map under the same geometry: the border points no longer lie on it
    h.set_map(grid[:20, :30].copy(), origin, res)
    with pytest.raises(mpmpc.MpmpcError, match="outside the map"):
        h.build_corridor(5, 0.02, 0.01)
    h.close()


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
    # BIT EXACT against the reference's own tables: the integer work (rasterisation, segment scan) is exact, the
    # per-waypoint trigonometry is computed once by the host's libm (mpmpc_set_path_geometry), the side of a border
    # is a cross product, and what is left on the device is +, -, x, sqrt - all correctly rounded
    assert np.array_equal(ub, g3["ub_" + key]) and np.array_equal(lb, g3["lb_" + key])
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
        assert bad == 0 and np.array_equal(ub_d, ub_h) and np.array_equal(lb_d, lb_h)
        m.add_obstacles([Obstacle(cx=-0.3, cy=-1.0, radius=0.08)])       # the map changes ...
        assert bm.update_corridor_from_map() == 0                        # ... and the table follows
    assert np.any(ub_d != np.load(M.GOLDEN + "/g3_corridor.npz")["ub_free"][:, :30])
    wp = np.arange(0, 160, 10).astype(np.int32)
    x0 = np.zeros((16, 3))
    u, plan, status, sol = bm.get_control_batch(wp, x0, np.zeros((16, 60)))
    assert np.all(np.isin(status, (1, 2, -3)))
