"""Kinematic bicycle model in path (spatial) coordinates - host side.

Mirror of the reference's src/spatial_bicycle_models.py: `TemporalState`, `SimpleSpatialState`,
`SpatialBicycleModel`, `BicycleModel(reference_path, length, width, Ts)` with `linearize`, `t2s`,
`s2t`, `drive`, `get_current_waypoint`, attributes `s, wp_id, current_waypoint, spatial_state,
temporal_state, length, width, safety_margin, Ts, n_states, reference_path`.

`linearize` (src/spatial_bicycle_models.py:391-417) is what K1 evaluates per stage on the GPU; the
Python method is kept with the same signature and evaluation order for single calls and tests.
Batched helpers `t2s_batch` / `current_waypoint_batch` serve the batched controller.
"""
from __future__ import annotations

import math

import numpy as np

CAR = '#F1C40F'
CAR_OUTLINE = '#B7950B'


class _StateVector:
    """Attribute bag addressable by position: v[i], v[a:b] (-> list), v[i] = x, len(v), v += array."""
    members: tuple = ()

    def __getitem__(self, item):
        names = [self.members[item]] if isinstance(item, int) else self.members[item]
        return [getattr(self, n) for n in names]

    def __setitem__(self, key, value):
        setattr(self, self.members[key], value)

    def __len__(self):
        return len(self.members)

    def __iadd__(self, other):
        for i, name in enumerate(self.members):
            setattr(self, name, getattr(self, name) + other[i])
        return self

    def list_states(self):
        return self.members


class TemporalState(_StateVector):
    """Pose in the world frame: x [m], y [m], psi [rad]."""

    def __init__(self, x, y, psi):
        self.members = ['x', 'y', 'psi']
        self.x, self.y, self.psi = x, y, psi


class SpatialState(_StateVector):
    """Base of path-relative states."""


class SimpleSpatialState(SpatialState):
    """Lateral offset e_y [m], heading error e_psi [rad], time t [s]."""

    def __init__(self, e_y=0.0, e_psi=0.0, t=0.0):
        self.members = ['e_y', 'e_psi', 't']
        self.e_y, self.e_psi, self.t = e_y, e_psi, t


def t2s_batch(x, y, psi, wx, wy, wpsi):
    """Vectorised src/spatial_bicycle_models.py:183-219: poses -> (e_y, e_psi, t=0) per instance."""
    e_y = np.cos(wpsi) * (y - wy) - np.sin(wpsi) * (x - wx)
    e_psi = np.mod(psi - wpsi + math.pi, 2 * math.pi) - math.pi
    return np.stack([e_y, e_psi, np.zeros_like(e_y)], axis=-1)


def current_waypoint_batch(segment_lengths, s):
    """Vectorised src/spatial_bicycle_models.py:256-279 (closest of the two enclosing waypoints,
    ties to the earlier one)."""
    cum = np.cumsum(segment_lengths)
    s = np.asarray(s, float)
    nxt = np.searchsorted(cum, s, side='right')
    prv = nxt - 1
    take_next = np.abs(s - cum[nxt]) < np.abs(s - cum[prv])
    return np.where(take_next, nxt, prv)


class SpatialBicycleModel:
    def __init__(self, reference_path, length, width, Ts):
        self.eps = 1e-12
        self.length = length
        self.width = width
        self.safety_margin = self._compute_safety_margin()
        self.reference_path = reference_path
        self.s = 0.0
        self.Ts = Ts
        self.wp_id = 0
        self.current_waypoint = self.reference_path.waypoints[self.wp_id]
        self.spatial_state = None
        self.temporal_state = None

    def _compute_safety_margin(self):
        return self.width / np.sqrt(2)

    def s2t(self, reference_waypoint, reference_state):
        w = reference_waypoint
        if isinstance(reference_state, np.ndarray):
            e_y, e_psi = reference_state[0], reference_state[1]
        elif isinstance(reference_state, SpatialState):
            e_y, e_psi = reference_state.e_y, reference_state.e_psi
        else:
            print('Reference State type not supported!')
            raise SystemExit(1)
        return TemporalState(w.x - e_y * np.sin(w.psi), w.y + e_y * np.cos(w.psi), w.psi + e_psi)

    def t2s(self, reference_waypoint, reference_state):
        w = reference_waypoint
        if isinstance(reference_state, np.ndarray):
            px, py, ppsi = reference_state[0], reference_state[1], reference_state[2]
        elif isinstance(reference_state, TemporalState):
            px, py, ppsi = reference_state.x, reference_state.y, reference_state.psi
        else:
            print('Reference State type not supported!')
            raise SystemExit(1)
        e_y = np.cos(w.psi) * (py - w.y) - np.sin(w.psi) * (px - w.x)
        e_psi = np.mod(ppsi - w.psi + math.pi, 2 * math.pi) - math.pi
        return SimpleSpatialState(e_y, e_psi, 0.0)

    def drive(self, u):
        """One forward-Euler step of the plant, then the arc-length update from the PRE-step
        spatial state (src/spatial_bicycle_models.py:221-244)."""
        v, delta = u
        psi = self.temporal_state.psi
        rates = np.array([v * np.cos(psi), v * np.sin(psi), v / self.length * np.tan(delta)])
        self.temporal_state += rates * self.Ts
        s_dot = 1 / (1 - self.spatial_state.e_y * self.current_waypoint.kappa) * v * np.cos(self.spatial_state.e_psi)
        self.s += s_dot * self.Ts

    def get_current_waypoint(self):
        cum = np.cumsum(self.reference_path.segment_lengths)
        nxt = (cum > self.s).searchsorted(True)
        prv = nxt - 1
        pick = nxt if np.abs(self.s - cum[nxt]) < np.abs(self.s - cum[prv]) else prv
        self.wp_id = pick
        self.current_waypoint = self.reference_path.waypoints[pick]

    def show(self):
        import matplotlib.patches as patches
        import matplotlib.pyplot as plt
        st = self.temporal_state
        c, s_ = np.cos(st.psi), np.sin(st.psi)
        corner = (st.x - (self.length / 2 * c - self.width / 2 * s_), st.y - (self.width / 2 * c + self.length / 2 * s_))
        plt.gca().add_patch(patches.Rectangle(corner, width=self.length, height=self.width, angle=np.rad2deg(st.psi),
                                              facecolor=CAR, edgecolor=CAR_OUTLINE, zorder=20))

    def get_spatial_derivatives(self, state, input, kappa):
        raise NotImplementedError

    def linearize(self, v_ref, kappa_ref, delta_s):
        raise NotImplementedError


class BicycleModel(SpatialBicycleModel):
    def __init__(self, reference_path, length, width, Ts):
        super().__init__(reference_path, length=length, width=width, Ts=Ts)
        self.spatial_state = SimpleSpatialState()
        self.n_states = len(self.spatial_state)
        self.temporal_state = self.s2t(reference_state=self.spatial_state, reference_waypoint=self.current_waypoint)

    def get_temporal_derivatives(self, state, input, kappa):
        e_y, e_psi, _ = state
        v, delta = input
        return 1 / (1 - (e_y * kappa)) * v * np.cos(e_psi), v / self.length * np.tan(delta)

    def get_spatial_derivatives(self, state, input, kappa):
        e_y, e_psi, _ = state
        v, _ = input
        s_dot, psi_dot = self.get_temporal_derivatives(state, input, kappa)
        return np.array([v * np.sin(e_psi) / s_dot, psi_dot / s_dot - kappa, 1 / s_dot])

    def linearize(self, v_ref, kappa_ref, delta_s):
        """f, A, B of x_{k+1} = A x_k + B u_k + (f - B u_ref); same operation order as the reference."""
        A = np.array([[1, delta_s, 0],
                      [-kappa_ref ** 2 * delta_s, 1, 0],
                      [-kappa_ref / v_ref * delta_s, 0, 1]])
        B = np.array([[0, 0],
                      [0, delta_s],
                      [-1 / (v_ref ** 2) * delta_s, 0]])
        f = np.array([0.0, 0.0, 1 / v_ref * delta_s])
        return f, A, B
