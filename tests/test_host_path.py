"""Host-side mirror of the reference's path / map / model classes against golden data that the
reference itself produced (tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import pytest

import mpc_np as M
import mpmpc_testlib as tl
from map import Map, Obstacle, fill_small_holes, line_aa
from reference_path import ReferencePath, Waypoint
from spatial_bicycle_models import BicycleModel, SimpleSpatialState, TemporalState, current_waypoint_batch, t2s_batch

G = M.GOLDEN


def _grid(g1, key):
    h, w = g1["grid_shape"]
    return np.unpackbits(g1[key])[:h * w].reshape(h, w).astype(np.int8)


@pytest.fixture(scope="module")
def g1():
    return np.load(G + "/g1_path_sim_track.npz")


@pytest.fixture(scope="module")
def sim_path(g1):
    m = Map.from_grid(_grid(g1, "grid_free"), origin=[-1, -2], resolution=0.005)
    rp = ReferencePath(m, list(g1["wp_x"]), list(g1["wp_y"]), 0.05, smoothing_distance=5, max_width=0.23,
                       circular=True)
    return m, rp


def test_line_aa_matches_skimage_sequences():
    g = np.load(G + "/g7_line_aa.npz")
    for i, (r0, c0, r1, c1) in enumerate(g["ends"]):
        rr, cc, _ = line_aa(int(r0), int(c0), int(r1), int(c1))
        lo, hi = g["ptr"][i], g["ptr"][i + 1]
        assert np.array_equal(rr, g["rr"][lo:hi]) and np.array_equal(cc, g["cc"][lo:hi]), i


def test_hole_removal_matches_reference_grid(g1):
    thr = _grid(g1, "grid_thresholded").astype(bool)
    assert np.array_equal(fill_small_holes(thr, 5).astype(np.int8), _grid(g1, "grid_free"))


def test_obstacle_stamping_matches_reference_grid(g1):
    m = Map.from_grid(_grid(g1, "grid_free"), origin=[-1, -2], resolution=0.005)
    m.add_obstacles([Obstacle(cx=c[0], cy=c[1], radius=c[2]) for c in g1["obstacles"]])
    assert np.array_equal(m.data, _grid(g1, "grid_obstacles"))
    assert m.w2m(0.0, 0.0) == (200, 400) and m.m2w(200, 400) == ((200.5) * 0.005 - 1, (400.5) * 0.005 - 2)


def test_path_construction_matches_reference(g1, sim_path):
    m, rp = sim_path
    assert rp.n_waypoints == 200
    for key, got in (("x", [w.x for w in rp.waypoints]), ("y", [w.y for w in rp.waypoints]),
                     ("psi", [w.psi for w in rp.waypoints]), ("kappa", [float(w.kappa) for w in rp.waypoints]),
                     ("segment_lengths", rp.segment_lengths), ("ub_static", [w.ub for w in rp.waypoints]),
                     ("lb_static", [w.lb for w in rp.waypoints])):
        # x, y, segment lengths are bit exact; psi goes through arctan2 (1 ulp between the numpy that
        # made the fixture and the one running here) and kappa divides that by ~0.05
        tol = {"psi": 1e-15, "kappa": 1e-13}.get(key, 0.0 if key in ("x", "y", "segment_lengths") else 1e-14)
        assert np.max(np.abs(np.array(got, float) - g1[key])) <= tol, key
    assert isinstance(rp.waypoints[0].kappa, int) and rp.waypoints[0].kappa == 0     # reference_path.py:182
    assert abs(rp.length - g1["length"][0]) < 1e-12
    assert np.allclose(np.array([w.static_border_cells[0] for w in rp.waypoints]), g1["border_ub"], atol=1e-14)
    assert np.allclose(np.array([w.static_border_cells[1] for w in rp.waypoints]), g1["border_lb"], atol=1e-14)
    kappa, v_ref, ds = rp.tables()
    assert np.allclose(ds, g1["ds_next"], rtol=0, atol=1e-15)
    assert isinstance(rp.get_waypoint(200 + 3), Waypoint) and rp.get_waypoint(203) is rp.waypoints[3]


def test_speed_profile_matches_certified_reference_qp(sim_path):
    m, rp = sim_path
    g2 = np.load(G + "/g2_speed_profile.npz")
    cons = dict(zip(('a_min', 'a_max', 'v_min', 'v_max', 'ay_max'), g2["constraints"]))
    P, q, A, l, u = rp.speed_profile_qp(cons)
    from scipy import sparse
    Aref = sparse.coo_matrix((g2["A_val"], (g2["A_row"], g2["A_col"])), shape=tuple(g2["A_shape"])).toarray()
    assert np.allclose(A, Aref, rtol=0, atol=1e-12) and np.allclose(q, g2["q"], atol=1e-14)
    assert np.allclose(l, g2["l"]) and np.allclose(u, g2["u"], atol=1e-14)
    # the device kernel's code, run through its CPU emulation (the product call has no host solver)
    rp.compute_speed_profile(cons, solver=tl.emu_speed_profile)
    v = np.array([w.v_ref for w in rp.waypoints])
    assert np.max(np.abs(v - g2["v_ref"])) < 1e-9


def test_speed_profile_kernel_code_against_dense_oracle():
    """K4's algorithm (emulated) against the oracle's dense certified QP solve on perturbed paths:
    other limits, tighter curvature caps, acceleration-limited stretches."""
    import osqp_np
    g1 = np.load(G + "/g1_path_sim_track.npz")
    rng = np.random.default_rng(11)
    for trial in range(6):
        n = int(rng.integers(20, 199))
        li = np.ascontiguousarray(g1["ds_next"][:n] * rng.uniform(0.7, 1.4, n))
        kappa = np.ascontiguousarray(g1["kappa"][:n].astype(float) * rng.uniform(0.5, 3.0))
        lim = np.array([-rng.uniform(0.05, 0.5), rng.uniform(0.1, 1.0), 0.0, rng.uniform(0.6, 1.5), rng.uniform(1.0, 5.0)])
        v, status, iters = tl.emu_speed_profile(li, kappa, lim)
        assert status[0] == 1 and 3 < iters[0] < 40
        vmax = np.minimum(lim[3], np.sqrt(lim[4] / (np.abs(kappa) + 1e-12)))
        D1 = np.zeros((n - 1, n))
        for i in range(n - 1):
            D1[i, i], D1[i, i + 1] = -1 / (2 * li[i]), 1 / (2 * li[i])
        A = np.vstack([D1, np.eye(n)])
        l = np.hstack([np.full(n - 1, lim[0]), np.full(n, lim[2])])
        u = np.hstack([np.full(n - 1, lim[1]), vmax])
        r = osqp_np.solve(np.eye(n), -vmax, A, l, u, osqp_np.Settings(polish=2, max_iter=20000))
        assert r.status == 1
        assert np.max(np.abs(v[0] - r.x)) < 1e-8, trial
    # inconsistent limits are refused, not solved
    _, status, _ = tl.emu_speed_profile(g1["ds_next"][:10], g1["kappa"][:10].astype(float), [0.5, -0.1, 0, 1, 4])
    assert status[0] == -1


@pytest.mark.parametrize("obst", [False, True])
def test_corridor_tables_match_reference(obst, g1, sim_path):
    """update_path_constraints for every start waypoint: G3."""
    g3 = np.load(G + "/g3_corridor.npz")
    m = Map.from_grid(_grid(g1, "grid_obstacles" if obst else "grid_free"), origin=[-1, -2], resolution=0.005)
    _, rp0 = sim_path
    rp = ReferencePath.from_tables(m, g1["x"], g1["y"], g1["psi"], g1["kappa"], circular=True,
                                   border_ub=g1["border_ub"], border_lb=g1["border_lb"])
    sm = float(g3["safety_margin"][0])
    ub, lb = rp.corridor_table(50, 2 * sm, sm)
    key = "obstacles" if obst else "free"
    assert np.allclose(ub, g3["ub_" + key], rtol=0, atol=1e-13, equal_nan=True)
    assert np.allclose(lb, g3["lb_" + key], rtol=0, atol=1e-13, equal_nan=True)
    u30, l30, cells = rp.update_path_constraints(17, 30, 2 * sm, sm)
    assert np.array_equal(u30, ub[16, :30]) and len(cells) == 30


def test_model_helpers(sim_path):
    m, rp = sim_path
    car = BicycleModel(reference_path=rp, length=0.12, width=0.06, Ts=0.05)
    assert car.n_states == 3 and abs(car.safety_margin - 0.06 / np.sqrt(2)) < 1e-18
    f, A, B = car.linearize(0.9, 2.0, 0.05)
    assert A.shape == (3, 3) and B.shape == (3, 2) and f.shape == (3,)
    assert A[1, 0] == -2.0 ** 2 * 0.05 and A[2, 0] == -2.0 / 0.9 * 0.05 and B[2, 0] == -1 / (0.9 ** 2) * 0.05
    assert f[2] == 1 / 0.9 * 0.05 and A[0, 1] == 0.05 and B[1, 1] == 0.05
    st = SimpleSpatialState(0.1, 0.2, 0.3)
    assert st[:] == [0.1, 0.2, 0.3] and st[1] == [0.2] and len(st) == 3
    st += np.array([1.0, 1.0, 1.0])
    assert st.e_y == 1.1
    # waypoint localisation and t2s, scalar vs batched
    g = np.load(G + "/g4_assembly_N30.npz")
    wp = current_waypoint_batch(rp.segment_lengths, g["s"])
    assert np.array_equal(wp, g["wp_id"])
    for c in range(4):
        car.s = float(g["s"][c])
        car.get_current_waypoint()
        assert car.wp_id == g["wp_id"][c]
        car.temporal_state = TemporalState(*g["pose"][c])
        s = car.t2s(reference_state=car.temporal_state, reference_waypoint=car.current_waypoint)
        assert np.allclose(s[:], g["x0"][c], atol=1e-15)
    w = g["wp_id"]
    x0 = t2s_batch(g["pose"][:, 0], g["pose"][:, 1], g["pose"][:, 2],
                   np.array([p.x for p in rp.waypoints])[w], np.array([p.y for p in rp.waypoints])[w],
                   np.array([p.psi for p in rp.waypoints])[w])
    assert np.allclose(x0, g["x0"], atol=1e-15)
