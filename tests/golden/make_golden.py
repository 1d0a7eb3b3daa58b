"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Run in the authoring container only (the reference cannot travel to the GPU box):

    cd /root/reference/src && MPLBACKEND=Agg /opt/conda/bin/python3.9 -W ignore \
        /root/repo/tests/golden/make_golden.py [stage ...]

Needs the conda interpreter (skimage, scipy<1.14 for `.A`) and
cwd = /root/reference/src (the reference loads `maps/sim_map.png` relatively and uses
flat imports).  `osqp` resolves to tests/golden/standin/osqp.py.

Everything written is DATA (inputs and the reference's outputs on them); no reference
source text is stored.

Stages
  path      G1  Sim_Track waypoint table, segment lengths, static borders, both grids
  speed     G2  speed-profile QP capture (reference_path.py:289-354) + certified v_ref
  corridor  G3  update_path_constraints tables [200 x 50], with and without obstacles
  assembly  G4  exact (P,q,A,l,u) the reference hands to osqp.setup for seeded cases
  assembly_full  G4f  the same with non-diagonal Q, R, QN (g4f_assembly_N*.npz: P as triplets)
  loop      G6  closed-loop lap, N=10, teacher-forced per-step record
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "standin"), "/root/reference/src"]
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np  # noqa: E402
from scipy import sparse  # noqa: E402

import osqp  # noqa: E402  (the stand-in)
from map import Map, Obstacle  # noqa: E402
from reference_path import ReferencePath  # noqa: E402
from spatial_bicycle_models import BicycleModel  # noqa: E402
from MPC import MPC  # noqa: E402

WP_X = [-0.75, -0.25, -0.25, 0.25, 0.25, 1.25, 1.25, 0.75, 0.75, 1.25, 1.25, -0.75, -0.75, -0.25]
WP_Y = [-1.5, -1.5, -0.5, -0.5, -1.5, -1.5, -1, -1, -0.5, -0.5, 0, 0, -1.5, -1.5]
OBSTACLES = [(0.0, 0.0, 0.05), (-0.8, -0.5, 0.08), (-0.7, -1.5, 0.05), (-0.3, -1.0, 0.08),
             (0.27, -1.0, 0.05), (0.78, -1.47, 0.05), (0.73, -0.9, 0.07), (1.2, 0.0, 0.08),
             (0.67, -0.05, 0.06)]
NMAX = 50
# one non-diagonal weight set (round 5): symmetric positive definite Q, R, QN with every off-diagonal entry set
FULL_WEIGHTS = dict(Q=[[1.0, 0.2, 0.05], [0.2, 0.3, -0.1], [0.05, -0.1, 0.2]], R=[[0.5, 0.1], [0.1, 0.2]],
                    QN=[[1.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 0.4]])
SPEED = {'a_min': -0.1, 'a_max': 0.5, 'v_min': 0.0, 'v_max': 1.0, 'ay_max': 4.0}
CAR = dict(length=0.12, width=0.06, Ts=0.05)


def build_track():
    """Sim_Track exactly as src/simulation.py:20-35 builds it."""
    m = Map(file_path='maps/sim_map.png', origin=[-1, -2], resolution=0.005)
    rp = ReferencePath(m, WP_X, WP_Y, 0.05, smoothing_distance=5, max_width=0.23, circular=True)
    return m, rp


def add_obstacles(m):
    m.add_obstacles([Obstacle(cx=c[0], cy=c[1], radius=c[2]) for c in OBSTACLES])


def path_table(rp):
    wps = rp.waypoints
    n = len(wps)
    ds_next = np.array([rp.get_waypoint(i + 1) - rp.get_waypoint(i) for i in range(n)])
    return dict(
        x=np.array([w.x for w in wps], float), y=np.array([w.y for w in wps], float),
        psi=np.array([w.psi for w in wps], float), kappa=np.array([float(w.kappa) for w in wps]),
        kappa0_is_int=np.array([isinstance(wps[0].kappa, int)]),
        ds_next=ds_next, segment_lengths=np.array(rp.segment_lengths, float),
        length=np.array([rp.length]), lb_static=np.array([w.lb for w in wps], float),
        ub_static=np.array([w.ub for w in wps], float),
        border_ub=np.array([w.static_border_cells[0] for w in wps], float),
        border_lb=np.array([w.static_border_cells[1] for w in wps], float))


def corridor_table(rp, sm):
    n = rp.n_waypoints
    ub = np.full((n, NMAX), np.nan)
    lb = np.full((n, NMAX), np.nan)
    for wp_id in range(n):
        try:
            u, l, _ = rp.update_path_constraints(wp_id + 1, NMAX, 2 * sm, sm)
            ub[wp_id], lb[wp_id] = u, l
        except ValueError:      # max([]) at reference_path.py:547: no free segment
            pass
    return ub, lb


def stage_path():
    m, rp = build_track()
    t = path_table(rp)
    from PIL import Image
    raw = np.array(Image.open('maps/sim_map.png'))[:, :, 0]
    t["grid_thresholded"] = np.packbits((raw >= m.threshold_occupied).astype(np.uint8))   # before hole removal
    grid_free = np.packbits(m.data.astype(np.uint8))
    add_obstacles(m)
    grid_obs = np.packbits(m.data.astype(np.uint8))
    np.savez_compressed(os.path.join(HERE, "g1_path_sim_track.npz"), grid_shape=np.array(m.data.shape),
                        grid_free=grid_free, grid_obstacles=grid_obs,
                        origin=np.array([-1.0, -2.0]), resolution=np.array([0.005]),
                        wp_x=np.array(WP_X), wp_y=np.array(WP_Y), obstacles=np.array(OBSTACLES), **t)
    print("G1:", {k: v.shape for k, v in t.items()})


def stage_speed():
    m, rp = build_track()
    osqp.CAPTURES.clear()
    rp.compute_speed_profile(dict(SPEED))
    cap = osqp.CAPTURES[-1]
    res = cap["res"]
    v_ref = np.array([w.v_ref for w in rp.waypoints], float)
    A = cap["A"].tocoo()
    np.savez_compressed(os.path.join(HERE, "g2_speed_profile.npz"),
                        P_diag=cap["P"].diagonal(), q=cap["q"], l=cap["l"], u=cap["u"],
                        A_row=A.row, A_col=A.col, A_val=A.data, A_shape=np.array(A.shape),
                        x=res.x, y=res.y, status=np.array([res.status]), v_ref=v_ref,
                        constraints=np.array([SPEED[k] for k in ('a_min', 'a_max', 'v_min', 'v_max', 'ay_max')]))
    print("G2: status", res.status, "iters", res.iters, "polished", res.polished,
          "v_ref range", v_ref.min(), v_ref.max())


def stage_corridor():
    m, rp = build_track()
    car = BicycleModel(reference_path=rp, **CAR)
    sm = car.safety_margin
    ub0, lb0 = corridor_table(rp, sm)
    add_obstacles(m)
    ub1, lb1 = corridor_table(rp, sm)
    np.savez_compressed(os.path.join(HERE, "g3_corridor.npz"), ub_free=ub0, lb_free=lb0,
                        ub_obstacles=ub1, lb_obstacles=lb1, safety_margin=np.array([sm]))
    print("G3: free ub", np.nanmin(ub0), np.nanmax(ub0), "lb", np.nanmin(lb0), np.nanmax(lb0),
          "| obst ub", np.nanmin(ub1), np.nanmax(ub1), "nan rows", np.isnan(ub1[:, 0]).sum())


def make_controller(rp, N, weights):
    car = BicycleModel(reference_path=rp, **CAR)
    if weights == "stock":            # simulation.py:101-103
        Q, R, QN = sparse.diags([1.0, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([1.0, 0.0, 0.0])
    elif weights == "full":           # NON-diagonal Q, R, QN (round 5): the reference accepts any sparse matrices (src/MPC.py:150)
        Q, R, QN = (sparse.csc_matrix(np.array(FULL_WEIGHTS[k])) for k in ("Q", "R", "QN"))
    else:                             # config 3 "time optimal" (build-defined)
        Q, R, QN = sparse.diags([0.3, 0.0, 0.0]), sparse.diags([0.5, 0.0]), sparse.diags([0.3, 0.0, 1.0])
    ic = {'umin': np.array([0.0, -np.tan(0.66) / car.length]),
          'umax': np.array([1.0, np.tan(0.66) / car.length])}
    sc = {'xmin': np.array([-np.inf, -np.inf, -np.inf]), 'xmax': np.array([np.inf, np.inf, np.inf])}
    return car, MPC(car, N, Q, R, QN, sc, ic, 4.0)


def stage_assembly(cases=((3, "stock", 32), (10, "stock", 64), (30, "stock", 96), (50, "time_optimal", 64)), tag=""):
    """G4: what the reference passes to osqp.setup for seeded (s, pose, previous plan) cases."""
    osqp.SOLVE = False
    worlds = {}
    for obst in (False, True):
        m, rp = build_track()
        rp.compute_speed_profile_done = False
        if obst:
            add_obstacles(m)
        worlds[obst] = rp
    # v_ref: certified speed profile from G2 (identical for both worlds: computed before obstacles)
    v_ref = np.load(os.path.join(HERE, "g2_speed_profile.npz"))["v_ref"]
    for rp in worlds.values():
        for w, v in zip(rp.waypoints, v_ref):
            w.v_ref = v
    # 256 captures in all; the first 16 / 24 / 32 / 24 of each horizon are the cases of round 1
    for N, weights, ncase in cases:
        rng = np.random.default_rng(1000 + N)
        rec = {k: [] for k in ("s", "pose", "cc_prev", "obst", "wp_id", "x0", "lb", "ub", "q", "l", "u",
                               "P_diag", "P_nnz", "A_indptr", "A_indices", "A_data", "P_row", "P_col", "P_val")}
        for c in range(ncase):
            obst = bool(c % 2)
            rp = worlds[obst]
            car, mpc = make_controller(rp, N, weights)
            cum = np.cumsum(rp.segment_lengths)
            # edge cases first: wp 0 (int kappa), the 199->0 wrap, then random
            wp = [0, 199, 198, 200 - N if N < 200 else 0][c] if c < 4 else int(rng.integers(0, 200))
            wp = max(wp, 0)
            ds = rp.get_waypoint(wp + 1) - rp.get_waypoint(wp)
            s = cum[wp] + (rng.uniform(-0.45, 0.45) * ds if (c >= 4 and wp > 0) else 0.0)
            s = min(max(s, 0.0), rp.length - 1e-9)
            car.s = s
            w = rp.waypoints[wp]
            e_y, e_psi = rng.uniform(-0.02, 0.02), rng.uniform(-0.2, 0.2)
            car.temporal_state.x = w.x - e_y * np.sin(w.psi)
            car.temporal_state.y = w.y + e_y * np.cos(w.psi)
            car.temporal_state.psi = w.psi + e_psi
            cc = np.zeros(2 * N)
            if c % 4 >= 2:            # warm previous plan: exercises the kappa_pred speed cap
                cc[0::2] = [rp.get_waypoint(wp + k).v_ref for k in range(N)]
                cc[1::2] = rng.uniform(-0.3, 0.3, N)
            mpc.current_control = cc.copy()
            osqp.CAPTURES.clear()
            try:
                mpc.get_control()
            except ValueError:
                continue
            cap = osqp.CAPTURES[-1]
            ub, lb, _ = rp.update_path_constraints(car.wp_id + 1, N, 2 * car.safety_margin, car.safety_margin)
            P, A = cap["P"], cap["A"]
            A.sort_indices()
            rec["s"].append(s)
            rec["pose"].append([car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi])
            rec["cc_prev"].append(cc)
            rec["obst"].append(obst)
            rec["wp_id"].append(car.wp_id)
            rec["x0"].append(car.spatial_state[:])
            rec["lb"].append(lb)
            rec["ub"].append(ub)
            rec["q"].append(cap["q"])
            rec["l"].append(cap["l"])
            rec["u"].append(cap["u"])
            rec["P_diag"].append(P.diagonal())
            rec["P_nnz"].append(P.nnz)
            Pc = P.tocoo()            # the WHOLE matrix the reference handed over (the same pattern for every case of a file)
            rec["P_row"].append(Pc.row); rec["P_col"].append(Pc.col); rec["P_val"].append(Pc.data)
            rec["A_indptr"].append(A.indptr)
            rec["A_indices"].append(A.indices)
            rec["A_data"].append(A.data)
        a_ptr = np.cumsum([0] + [d.size for d in rec["A_data"]])
        out = {k: np.array(v) for k, v in rec.items() if not k.startswith("A_ind") and k != "A_data"}
        out.update(A_indptr=np.array(rec["A_indptr"]), A_indices=np.concatenate(rec["A_indices"]),
                   A_data=np.concatenate(rec["A_data"]), A_case_ptr=a_ptr, N=np.array([N]),
                   weights=np.array([weights]))
        if not tag:                   # (the files of rounds 1 - 4 keep their content: diagonal P, P_diag says it all)
            for k in ("P_row", "P_col", "P_val"):
                out.pop(k)
        np.savez_compressed(os.path.join(HERE, "g4%s_assembly_N%d.npz" % (tag, N)), **out)
        print("G4%s N=%d: %d cases, nnz(A) %s, nnz(P) %s, wp_ids %s" % (tag, N, len(rec["s"]), sorted(set(np.diff(a_ptr))),
                                                                   sorted(set(rec["P_nnz"])), rec["wp_id"][:8]))
    osqp.SOLVE = True


def stage_assembly_full():
    """G4f: the same captures with NON-diagonal Q, R, QN (FULL_WEIGHTS): the reference puts the whole matrices into P
    (src/MPC.py:150) and only diag(Q), diag(R) - but the whole QN - into q (src/MPC.py:153-155)."""
    stage_assembly(cases=((3, "full", 16), (10, "full", 32), (30, "full", 48), (50, "full", 24)), tag="f")


def stage_loop():
    """G6: closed-loop lap of src/simulation.py's while-loop (simulation.py:134-148), per-step record.

    Each step's inputs are recorded so a consumer can be teacher-forced: free-running
    trajectories of different solvers legitimately diverge (cost-free kappa_{N-1} feeds
    the kappa_pred speed cap of the next step).
    """
    for N in (10, 30):
        m, rp = build_track()
        add_obstacles(m)
        car, mpc = make_controller(rp, N, "stock")
        rp.compute_speed_profile(dict(SPEED))
        rec = {k: [] for k in ("s", "pose", "cc_prev", "wp_id", "x0", "lb", "ub", "status", "u", "counter",
                               "z", "cc_next", "pred_x", "pred_y")}
        exited = False
        while car.s < rp.length:
            s, pose = car.s, [car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi]
            cc_prev = mpc.current_control.copy()
            osqp.CAPTURES.clear()
            try:
                u = mpc.get_control()
            except SystemExit:      # src/MPC.py:218-220: N-1 consecutive infeasible steps end the run
                exited = True
                break
            cap = osqp.CAPTURES[-1]
            res = cap["res"]
            ub, lb, _ = rp.update_path_constraints(car.wp_id + 1, N, 2 * car.safety_margin, car.safety_margin)
            rec["s"].append(s)
            rec["pose"].append(pose)
            rec["cc_prev"].append(cc_prev)
            rec["wp_id"].append(car.wp_id)
            rec["x0"].append(car.spatial_state[:])
            rec["lb"].append(lb)
            rec["ub"].append(ub)
            rec["status"].append(res.status)
            rec["u"].append(np.array(u, float))
            rec["counter"].append(mpc.infeasibility_counter)
            rec["z"].append(res.x if res.status > 0 else np.full(5 * N + 3, np.nan))
            rec["cc_next"].append(mpc.current_control.copy())
            # MPC.update_prediction (src/MPC.py:224-248): what the reference stores for plotting after this step
            # (unchanged by an infeasible step: the previous prediction stays)
            px, py = mpc.current_prediction if mpc.current_prediction is not None else ([np.nan] * (N - 2), [np.nan] * (N - 2))
            rec["pred_x"].append(np.array(px, float))
            rec["pred_y"].append(np.array(py, float))
            car.drive(u)
        np.savez_compressed(os.path.join(HERE, "g6_closed_loop_N%d.npz" % N),
                            **{k: np.array(v) for k, v in rec.items()}, N=np.array([N]),
                            final_s=np.array([car.s]), exited=np.array([exited]))
        st = np.array(rec["status"])
        print("G6 N=%d: %d steps, infeasible %d, final s %.4f, exit(1) %s" % (N, st.size, (st < 0).sum(), car.s, exited))


def stage_loop_stock():
    """G6s: the SAME closed loop with the stand-in in STOCK mode - the restated OSQP at its defaults and nothing else
    (osqp_np.Settings(polish=0): eps 1e-3, no polish, no phase 1), i.e. the arithmetic of the reference's own solver call
    (src/MPC.py:159,183).  Recorded per step: the inputs (teacher forcing), OSQP's verdict and which branch of
    src/MPC.py:185-220 the reference took (fresh plan / fallback).  ADVICE r2: the default path of the build must take
    the same branch (tests/test_emul_parity.py, tests/test_gpu_parity.py)."""
    import osqp_np
    saved = osqp.SETTINGS
    osqp.SETTINGS = osqp_np.Settings(polish=0)
    try:
        for N in (10, 30):
            m, rp = build_track()
            add_obstacles(m)
            car, mpc = make_controller(rp, N, "stock")
            rp.compute_speed_profile(dict(SPEED))
            rec = {k: [] for k in ("s", "pose", "cc_prev", "wp_id", "x0", "lb", "ub", "status", "iters", "u", "counter", "pri_res")}
            exited = False
            while car.s < rp.length:
                s, pose = car.s, [car.temporal_state.x, car.temporal_state.y, car.temporal_state.psi]
                cc_prev = mpc.current_control.copy()
                osqp.CAPTURES.clear()
                try:
                    u = mpc.get_control()
                except SystemExit:
                    exited = True
                    break
                res = osqp.CAPTURES[-1]["res"]
                ub, lb, _ = rp.update_path_constraints(car.wp_id + 1, N, 2 * car.safety_margin, car.safety_margin)
                for k, v in (("s", s), ("pose", pose), ("cc_prev", cc_prev), ("wp_id", car.wp_id), ("x0", car.spatial_state[:]),
                             ("lb", lb), ("ub", ub), ("status", res.status), ("iters", res.iters), ("u", np.array(u, float)),
                             ("counter", mpc.infeasibility_counter), ("pri_res", res.pri_res)):
                    rec[k].append(v)
                car.drive(u)
            np.savez_compressed(os.path.join(HERE, "g6s_stock_loop_N%d.npz" % N), **{k: np.array(v) for k, v in rec.items()},
                                N=np.array([N]), final_s=np.array([car.s]), exited=np.array([exited]))
            st = np.array(rec["status"])
            print("G6s N=%d: %d steps, statuses %s, final s %.4f, exit(1) %s" % (N, st.size, dict(zip(*np.unique(st, return_counts=True))), car.s, exited))
    finally:
        osqp.SETTINGS = saved


def stage_raster():
    """G7: skimage.draw.line_aa cell sequences (order matters to _compute_free_segments)."""
    from skimage.draw import line_aa
    rng = np.random.default_rng(7)
    ends = rng.integers(0, 500, (300, 4))
    ends[:20, 2:] = ends[:20, :2] + rng.integers(-3, 4, (20, 2))        # very short lines
    ends[20] = [5, 5, 5, 5]                                             # single cell
    ends[21] = [10, 3, 10, 40]                                          # axis aligned
    ends[22] = [3, 10, 40, 10]
    ends[23] = [0, 0, 30, 30]                                           # diagonal
    ends = np.clip(ends, 0, 499)
    rr, cc, ptr = [], [], [0]
    for r0, c0, r1, c1 in ends:
        a, b, _ = line_aa(int(r0), int(c0), int(r1), int(c1))
        rr.append(a)
        cc.append(b)
        ptr.append(ptr[-1] + a.size)
    np.savez_compressed(os.path.join(HERE, "g7_line_aa.npz"), ends=ends, rr=np.concatenate(rr),
                        cc=np.concatenate(cc), ptr=np.array(ptr))
    print("G7:", len(ends), "lines,", ptr[-1], "cells")


STAGES = dict(raster=stage_raster, path=stage_path, speed=stage_speed, corridor=stage_corridor, assembly=stage_assembly,
              loop=stage_loop, loop_stock=stage_loop_stock, assembly_full=stage_assembly_full)

if __name__ == "__main__":
    assert os.getcwd().rstrip("/") == "/root/reference/src", "run with cwd=/root/reference/src"
    for name in (sys.argv[1:] or list(STAGES)):
        STAGES[name]()
