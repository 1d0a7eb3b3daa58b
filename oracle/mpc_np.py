"""ORACLE (test infrastructure, never shipped, never on the product path).

numpy restatement of the QP-assembly half of the reference hot path
(/root/reference/src/MPC.py:61-155 and the helpers it reaches), written to reproduce
the reference's float64 results bit for bit.  Pinned by tests/golden/g4_assembly_*.npz,
which hold what the reference itself handed to `osqp.setup` (tests/golden/make_golden.py).

Decision vector  w = [x_0 .. x_N (e_y, e_psi, t each), u_0 .. u_{N-1} (v, kappa each)], n = 5N+3.
Rows             [3(N+1) dynamics equalities ; 3(N+1) state boxes ; 2N input boxes], m = 8N+6.
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


@dataclasses.dataclass
class Track:
    """Per-waypoint tables of one reference path (what `ReferencePath` holds)."""
    x: np.ndarray
    y: np.ndarray
    psi: np.ndarray
    kappa: np.ndarray
    ds_next: np.ndarray          # waypoint[i+1] - waypoint[i], circular (reference_path.py:50-57)
    segment_lengths: np.ndarray  # [0.0, d(0,1), d(1,2), ...]  (reference_path.py:201)
    v_ref: np.ndarray
    length: float
    circular: bool = True

    @property
    def n(self):
        return self.x.size

    @staticmethod
    def sim_track() -> "Track":
        g1 = np.load(os.path.join(GOLDEN, "g1_path_sim_track.npz"))
        g2 = np.load(os.path.join(GOLDEN, "g2_speed_profile.npz"))
        return Track(g1["x"], g1["y"], g1["psi"], g1["kappa"], g1["ds_next"],
                     g1["segment_lengths"], g2["v_ref"], float(g1["length"][0]), True)


@dataclasses.dataclass
class Weights:
    Q: np.ndarray                # 3x3 (only its diagonal enters q, MPC.py:153)
    R: np.ndarray                # 2x2 (only its diagonal enters q, MPC.py:155)
    QN: np.ndarray               # 3x3, used in full (MPC.py:150,154)

    @staticmethod
    def stock():                 # simulation.py:101-103
        return Weights(np.diag([1.0, 0.0, 0.0]), np.diag([0.5, 0.0]), np.diag([1.0, 0.0, 0.0]))

    @staticmethod
    def time_optimal():          # build-defined
        return Weights(np.diag([0.3, 0.0, 0.0]), np.diag([0.5, 0.0]), np.diag([0.3, 0.0, 1.0]))


@dataclasses.dataclass
class Limits:
    xmin: np.ndarray
    xmax: np.ndarray
    umin: np.ndarray
    umax: np.ndarray
    ay_max: float = 4.0
    length: float = 0.12         # wheelbase L

    @staticmethod
    def stock(length=0.12):      # simulation.py:105-111
        k = np.tan(0.66) / length
        return Limits(np.full(3, -np.inf), np.full(3, np.inf), np.array([0.0, -k]),
                      np.array([1.0, k]), 4.0, length)


# ---- a2: SpatialBicycleModel.get_current_waypoint (spatial_bicycle_models.py:256-279) -------
def current_waypoint(segment_lengths, s):
    cum = np.cumsum(segment_lengths)
    nxt = int((cum > s).searchsorted(True))
    prv = nxt - 1
    return nxt if abs(s - cum[nxt]) < abs(s - cum[prv]) else prv


# ---- a3: t2s (spatial_bicycle_models.py:183-219) -------------------------------------------
def t2s(x, y, psi, wx, wy, wpsi):
    e_y = np.cos(wpsi) * (y - wy) - np.sin(wpsi) * (x - wx)
    e_psi = np.mod(psi - wpsi + math.pi, 2 * math.pi) - math.pi
    return e_y, e_psi, 0.0


def s2t(wx, wy, wpsi, e_y, e_psi):
    return wx - e_y * np.sin(wpsi), wy + e_y * np.cos(wpsi), wpsi + e_psi


# ---- a5: linearize (spatial_bicycle_models.py:391-417), all stages at once -----------------
def linearize(v, kappa, ds):
    """Returns a10=A[1,0], a20=A[2,0], b20=B[2,0], f2=f[2]; A[0,1]=B[1,1]=ds, diag(A)=1."""
    a10 = (-(kappa * kappa)) * ds
    a20 = ((-kappa) / v) * ds
    b20 = ((-1.0) / (v * v)) * ds
    f2 = (1.0 / v) * ds
    return a10, a20, b20, f2


# ---- a7: kappa_pred quirk (MPC.py:86-87): broadcast add of the LAST element ---------------
def kappa_pred(cc, L):
    cc = np.asarray(cc, float)
    return np.tan(cc[3:] + cc[-1:]) / L


def stage_tables(track: Track, wp_id: int, N: int):
    idx = np.mod(wp_id + np.arange(N), track.n) if track.circular else wp_id + np.arange(N)
    return track.kappa[idx], track.v_ref[idx], track.ds_next[idx]


def assemble(track: Track, wp_id: int, x0, cc_prev, lb, ub, N: int, wts: Weights, lim: Limits):
    """Dense (Pdiag-or-P, q, A, l, u) exactly as MPC._init_problem builds them."""
    nx, nu = 3, 2
    kap, v, ds = stage_tables(track, wp_id, N)
    a10, a20, b20, f2 = linearize(v, kap, ds)
    n = nx * (N + 1) + nu * N
    m = nx * (N + 1) + n
    A = np.zeros((m, n))
    r = np.arange(nx * (N + 1))
    A[r, r] = -1.0
    for k in range(N):
        r0, c0, cu = nx * (k + 1), nx * k, nx * (N + 1) + nu * k
        A[r0 + 0, c0 + 0] = 1.0
        A[r0 + 0, c0 + 1] = ds[k]
        A[r0 + 1, c0 + 0] = a10[k]
        A[r0 + 1, c0 + 1] = 1.0
        A[r0 + 2, c0 + 0] = a20[k]
        A[r0 + 2, c0 + 2] = 1.0
        A[r0 + 1, cu + 1] = ds[k]
        A[r0 + 2, cu + 0] = b20[k]
    A[nx * (N + 1) + np.arange(n), np.arange(n)] = 1.0
    # offsets uq = B [v, kappa] - f   (MPC.py:107-108)
    uq = np.zeros(nx * N)
    uq[1::3] = ds * kap
    uq[2::3] = b20 * v - f2
    # dynamic speed cap (MPC.py:111-113)
    umax_dyn = np.tile(lim.umax, N).astype(float)
    kp = kappa_pred(cc_prev, lim.length)[:N]
    vmax = np.sqrt(lim.ay_max / (np.abs(kp) + 1e-12))
    umax_dyn[0::2] = np.where(vmax < umax_dyn[0::2], vmax, umax_dyn[0::2])
    xmin_dyn = np.tile(lim.xmin, N + 1).astype(float)
    xmax_dyn = np.tile(lim.xmax, N + 1).astype(float)
    xmin_dyn[0] = x0[0]
    xmax_dyn[0] = x0[0]
    xmin_dyn[nx::nx] = lb
    xmax_dyn[nx::nx] = ub
    xr = np.zeros(nx * (N + 1))
    xr[nx::nx] = (np.asarray(lb) + np.asarray(ub)) / 2
    ur = np.zeros(nu * N)
    ur[0::2] = v
    ur[1::2] = kap
    leq = np.hstack([-np.asarray(x0, float), uq])
    l = np.hstack([leq, xmin_dyn, np.tile(lim.umin, N)])
    u = np.hstack([leq, xmax_dyn, umax_dyn])
    P = np.zeros((n, n))
    for k in range(N):
        P[nx * k:nx * k + nx, nx * k:nx * k + nx] = wts.Q
        c = nx * (N + 1) + nu * k
        P[c:c + nu, c:c + nu] = wts.R
    P[nx * N:nx * N + nx, nx * N:nx * N + nx] = wts.QN
    q = np.hstack([-np.tile(np.diag(wts.Q), N) * xr[:-nx], -(wts.QN.dot(xr[-nx:])),
                   -np.tile(np.diag(wts.R), N) * ur])
    return P, q, A, l, u


def extract_control(z, N, L):
    """a12 (MPC.py:185-194): plan with odd entries converted kappa -> delta."""
    cc = np.array(z[-2 * N:], float)
    cc[1::2] = np.arctan(cc[1::2] * L)
    return cc
