"""Seeded synthetic workloads for the batched QP path.

Each scenario is a batch of independent controller instances on Sim_Track
(src/simulation.py:20-35 of the reference): a waypoint index, a pose near that waypoint
(turned into the spatial state the way MPC.get_control does, src/MPC.py:171-177) and a previous
plan `cc_prev` (MPC.current_control).  Track tables come from the committed fixtures that the
reference itself produced (tests/golden/make_golden.py).

This is the workload generator of bench.py, the tests and the profiling scripts: it lives beside bench.py, NOT in the
product package (multi-purpose-mpc_amd/ imports nothing from the test tree).
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")

CAR_LENGTH, CAR_WIDTH, TS = 0.12, 0.06, 0.05            # simulation.py:53-54
UMIN = np.array([0.0, -math.tan(0.66) / CAR_LENGTH])      # simulation.py:108-109
UMAX = np.array([1.0, math.tan(0.66) / CAR_LENGTH])
XMIN, XMAX = np.full(3, -np.inf), np.full(3, np.inf)      # simulation.py:110-111
AY_MAX = 4.0
WEIGHTS = {
    "stock": (np.array([1.0, 0.0, 0.0]), np.array([0.5, 0.0]), np.array([1.0, 0.0, 0.0])),  # simulation.py:101-103
    # the reference has no numeric time-optimal weights (README.md:56 is prose); build-defined:
    "time_optimal": (np.array([0.3, 0.0, 0.0]), np.array([0.5, 0.0]), np.array([0.3, 0.0, 1.0])),
}


# non-diagonal weight matrices (src/MPC.py:150 puts the whole Q, R, QN into the Hessian): one symmetric positive definite set
# with every off-diagonal entry set, and singular (rank-one) blocks - shared by the tests and the profiling scripts
QN_FULL = np.array([[1.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 0.4]])
Q_FULL = np.array([[1.0, 0.2, 0.05], [0.2, 0.3, -0.1], [0.05, -0.1, 0.2]])
R_FULL = np.array([[0.5, 0.1], [0.1, 0.2]])
Q_RANK1 = np.outer([1.0, 0.5, 0.0], [1.0, 0.5, 0.0])
R_RANK1 = np.outer([0.7, 0.02], [0.7, 0.02])
FULL_WEIGHT_SETS = {"full": (Q_FULL, R_FULL, QN_FULL), "q_only": (Q_FULL, np.diag([0.5, 0.0]), np.diag([1.0, 0.0, 0.0])),
                    "r_only": (np.diag([1.0, 0.0, 0.0]), R_FULL, np.diag([1.0, 0.0, 0.0])), "rank1": (Q_RANK1, R_RANK1, Q_RANK1)}


@dataclasses.dataclass
class Track:
    x: np.ndarray
    y: np.ndarray
    psi: np.ndarray
    kappa: np.ndarray
    ds_next: np.ndarray
    segment_lengths: np.ndarray
    v_ref: np.ndarray
    length: float
    ub_free: np.ndarray
    lb_free: np.ndarray
    ub_obstacles: np.ndarray
    lb_obstacles: np.ndarray

    @property
    def n_wp(self):
        return self.x.size


def sim_track(golden_dir: str = GOLDEN) -> Track:
    g1 = np.load(os.path.join(golden_dir, "g1_path_sim_track.npz"))
    g2 = np.load(os.path.join(golden_dir, "g2_speed_profile.npz"))
    g3 = np.load(os.path.join(golden_dir, "g3_corridor.npz"))
    return Track(g1["x"], g1["y"], g1["psi"], g1["kappa"], g1["ds_next"], g1["segment_lengths"],
                 g2["v_ref"], float(g1["length"][0]), g3["ub_free"], g3["lb_free"],
                 g3["ub_obstacles"], g3["lb_obstacles"])


@dataclasses.dataclass
class Scenario:
    name: str
    N: int
    weights: str
    obstacles: bool
    wp_id: np.ndarray      # [B] int32
    x0: np.ndarray         # [B,3]
    cc_prev: np.ndarray    # [B,2N]
    lb: np.ndarray         # [B,N] corridor rows (also available as table rows)
    ub: np.ndarray

    @property
    def B(self):
        return self.wp_id.size


CONFIGS = {
    2: dict(B=1024, N=30, weights="stock", obstacles=False, seed=2),
    3: dict(B=4096, N=50, weights="time_optimal", obstacles=False, seed=3),
    4: dict(B=8192, N=30, weights="stock", obstacles=True, seed=4),
    # config 5: 65 536 instances over 8 GPUs = contiguous shards of 8 192 per GPU
    5: dict(B=65536, B_per_gpu=8192, N=30, weights="stock", obstacles=True, seed=5),
}


def make(config: int, track: Track | None = None, B: int | None = None, N: int | None = None) -> Scenario:
    """Config `config` of BASELINE.json; B / N may be overridden for small parity cases."""
    tr = track or sim_track()
    spec = dict(CONFIGS[config])
    B = int(B or spec["B"])
    N = int(N or spec["N"])
    rng = np.random.default_rng(spec["seed"])
    ubT, lbT = (tr.ub_obstacles, tr.lb_obstacles) if spec["obstacles"] else (tr.ub_free, tr.lb_free)
    wp = rng.integers(0, tr.n_wp, B)
    if N > ubT.shape[1]:
        raise ValueError("the corridor tables of the golden track hold %d stages; N = %d needs its own tables" % (ubT.shape[1], N))
    ub, lb = ubT[wp, :N], lbT[wp, :N]
    u1, u2 = rng.uniform(-1.0, 1.0, B), rng.uniform(-0.2, 0.2, B)
    if spec["obstacles"]:      # offset inside the first horizon corridor: most, not all, feasible
        e_y = (lb[:, 0] + ub[:, 0]) / 2 + 0.3 * u1 * (ub[:, 0] - lb[:, 0]) / 2
    else:
        e_y = 0.02 * u1
    e_psi = u2
    # pose via s2t, then back through t2s exactly as get_control does
    wx, wy, wpsi = tr.x[wp], tr.y[wp], tr.psi[wp]
    px, py, ppsi = wx - e_y * np.sin(wpsi), wy + e_y * np.cos(wpsi), wpsi + e_psi
    x0 = np.stack([np.cos(wpsi) * (py - wy) - np.sin(wpsi) * (px - wx),
                   np.mod(ppsi - wpsi + math.pi, 2 * math.pi) - math.pi, np.zeros(B)], axis=1)
    delta = rng.uniform(-0.3, 0.3, (B, N))
    cc = np.zeros((B, 2 * N))
    warm = np.arange(B) >= B // 2          # first half cold (zeros), second half a warm previous plan
    idx = np.mod(wp[:, None] + np.arange(N)[None, :], tr.n_wp)
    cc[:, 0::2] = np.where(warm[:, None], tr.v_ref[idx], 0.0)
    cc[:, 1::2] = np.where(warm[:, None], delta, 0.0)
    return Scenario("config%d" % config, N, spec["weights"], spec["obstacles"], wp.astype(np.int32), x0, cc,
                    np.ascontiguousarray(lb), np.ascontiguousarray(ub))
