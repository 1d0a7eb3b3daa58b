"""Reference path: waypoints with heading, curvature, reference speed and drivable corridor.

Host-side mirror of the reference's src/reference_path.py (same class / method names, argument
order and return types, so src/simulation.py's body runs against it unchanged):

    Waypoint(x, y, psi, kappa)           src/reference_path.py:20-57   (a - b = euclidean distance)
    ReferencePath(map, wp_x, wp_y, resolution, smoothing_distance, max_width, circular)
        .waypoints .n_waypoints .length .segment_lengths .map .circular
        .get_waypoint(i)                 src/reference_path.py:356-371
        .compute_speed_profile(dict)     src/reference_path.py:289-354
        .update_path_constraints(wp_id, N, min_width, safety_margin) -> (ub, lb, cells)
                                         src/reference_path.py:522-648
        .show()

Additions used by the batched GPU path: `tables()` (per-waypoint kappa / v_ref / ds_next arrays for
`mpmpc_set_path`) and `corridor_table(n_cols, ...)` (update_path_constraints for every start
waypoint of a static map, for `mpmpc_set_corridor`); `ReferencePath.from_tables` rebuilds a path
from stored per-waypoint data (the committed fixtures) without the map image.

Faithfulness notes: curvature of waypoint 0 is the integer 0 (src/reference_path.py:182); the
forward projection of the previous corridor borders uses cos for both coordinates of the upper and
sin for both of the lower border (src/reference_path.py:559-562).  Both are reproduced, not fixed.
"""
from __future__ import annotations

import math

import numpy as np

from map import Map, Obstacle, line_aa  # noqa: F401  (Map / Obstacle re-exported like the reference)

DRIVABLE_AREA = '#BDC3C7'
WAYPOINTS = '#D0D3D4'
PATH_CONSTRAINTS = '#F5B041'
OBSTACLE = '#2E4053'


def _wrap(angle):
    return np.mod(angle + math.pi, 2 * math.pi) - math.pi


class Waypoint:
    __slots__ = ("x", "y", "psi", "kappa", "v_ref", "lb", "ub", "static_border_cells", "dynamic_border_cells")

    def __init__(self, x, y, psi, kappa):
        self.x, self.y, self.psi, self.kappa = x, y, psi, kappa
        self.v_ref = None
        self.lb = self.ub = None
        self.static_border_cells = None
        self.dynamic_border_cells = None

    def __sub__(self, other):
        return ((self.x - other.x) ** 2 + (self.y - other.y) ** 2) ** 0.5


class ReferencePath:
    def __init__(self, map, wp_x, wp_y, resolution, smoothing_distance, max_width, circular):
        self.eps = 1e-12
        self.map = map
        self.resolution = resolution
        self.smoothing_distance = smoothing_distance
        self.circular = circular
        self.waypoints = self._construct_path(wp_x, wp_y)
        self.n_waypoints = len(self.waypoints)
        self.length, self.segment_lengths = self._compute_length()
        self._compute_width(max_width=max_width)

    # ------------------------------------------------------------------ alternative constructor
    @classmethod
    def from_tables(cls, map, x, y, psi, kappa, circular=True, v_ref=None, border_ub=None, border_lb=None,
                    ub_static=None, lb_static=None, resolution=None, smoothing_distance=None):
        self = cls.__new__(cls)
        self.eps = 1e-12
        self.map = map
        self.resolution = resolution
        self.smoothing_distance = smoothing_distance
        self.circular = circular
        self.waypoints = [Waypoint(float(a), float(b), float(c), (0 if i == 0 and d == 0 else float(d)))
                          for i, (a, b, c, d) in enumerate(zip(x, y, psi, kappa))]
        self.n_waypoints = len(self.waypoints)
        self.length, self.segment_lengths = self._compute_length()
        for i, w in enumerate(self.waypoints):
            if v_ref is not None:
                w.v_ref = float(v_ref[i])
            if border_ub is not None:
                cells = (tuple(border_ub[i]), tuple(border_lb[i]))
                w.static_border_cells = cells
                w.dynamic_border_cells = cells
            if ub_static is not None:
                w.ub, w.lb = float(ub_static[i]), float(lb_static[i])
        return self

    # ------------------------------------------------------------------ construction
    def _construct_path(self, wp_x, wp_y):
        """Corner points -> evenly spaced points -> moving average -> Waypoint list
        (src/reference_path.py:110-193)."""
        counts = [int(np.sqrt((wp_x[i + 1] - wp_x[i]) ** 2 + (wp_y[i + 1] - wp_y[i]) ** 2) / self.resolution)
                  for i in range(len(wp_x) - 1)]
        xs, ys = [], []
        for i, cnt in enumerate(counts):
            xs.extend(np.linspace(wp_x[i], wp_x[i + 1], cnt, endpoint=False).tolist())
            ys.extend(np.linspace(wp_y[i], wp_y[i + 1], cnt, endpoint=False).tolist())
        xs.append(wp_x[-1])
        ys.append(wp_y[-1])
        sd = self.smoothing_distance
        centres = range(sd, len(xs) - sd)
        sx = [np.mean(xs[c - sd:c + sd + 1]) for c in centres]
        sy = [np.mean(ys[c - sd:c + sd + 1]) for c in centres]
        return self._construct_waypoints(list(zip(sx, sy)))

    def _construct_waypoints(self, waypoint_coordinates):
        coords = waypoint_coordinates
        out = []
        for i in range(len(coords) - 1):
            here, ahead = np.array(coords[i]), np.array(coords[i + 1])
            step = ahead - here
            psi = np.arctan2(step[1], step[0])
            dist_ahead = np.linalg.norm(step, 2)
            if i == 0:
                kappa = 0
            else:
                back = here - np.array(coords[i - 1])
                turn = np.mod(psi - np.arctan2(back[1], back[0]) + math.pi, 2 * math.pi) - math.pi
                kappa = turn / (dist_ahead + self.eps)
            out.append(Waypoint(here[0], here[1], psi, kappa))
        return out

    def _compute_length(self):
        seg = [0.0] + [self.waypoints[i + 1] - self.waypoints[i] for i in range(len(self.waypoints) - 1)]
        return sum(seg), seg

    def _compute_width(self, max_width):
        """Static corridor: walk from each waypoint towards both sides until an occupied cell
        (src/reference_path.py:206-287)."""
        for wp in self.waypoints:
            found = []
            for side in (+1, -1):                     # left, right
                ang = np.mod(wp.psi + side * math.pi / 2 + math.pi, 2 * math.pi) - math.pi
                tx, ty = self.map.w2m(wp.x + max_width * np.cos(ang), wp.y + max_width * np.sin(ang))
                found.append(self._get_min_width(wp, tx, ty, max_width))
            wp.ub = found[0][0]
            wp.lb = -1 * found[1][0]
            wp.static_border_cells = (found[0][1], found[1][1])
            wp.dynamic_border_cells = (found[0][1], found[1][1])

    def _get_min_width(self, wp, t_x, t_y, max_width):
        px, py = self.map.w2m(wp.x, wp.y)
        best, best_cell = max_width, None
        last = (t_x, t_y)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                last = (t_x + di, t_y + dj)
                cx, cy, _ = line_aa(px, py, last[0], last[1])
                for ix, iy in zip(cx, cy):
                    if self.map.data[iy, ix] == 0:
                        wx, wy = self.map.m2w(ix, iy)
                        dist = np.sqrt((wp.x - wx) ** 2 + (wp.y - wy) ** 2)
                        if dist < best:
                            best, best_cell = dist, (wx, wy)
        if best_cell is None:
            best_cell = self.map.m2w(last[0], last[1])    # no obstacle met: the last probed cell
        return best, best_cell

    # ------------------------------------------------------------------ speed profile
    def speed_profile_qp(self, Constraints):
        """(P, q, A, l, u) of the speed-profile problem, exactly as src/reference_path.py:297-344."""
        n = self.n_waypoints - 1
        a_min = np.ones(n - 1) * Constraints['a_min']
        a_max = np.ones(n - 1) * Constraints['a_max']
        v_min = np.ones(n) * Constraints['v_min']
        v_max = np.ones(n) * Constraints['v_max']
        D1 = np.zeros((n - 1, n))
        for i in range(n):
            li = self.get_waypoint(i + 1) - self.get_waypoint(i)
            ki = self.get_waypoint(i).kappa
            if i < n - 1:
                D1[i, i:i + 2] = np.array([-1 / (2 * li), 1 / (2 * li)])
            cap = np.sqrt(Constraints['ay_max'] / (np.abs(ki) + self.eps))
            if cap < v_max[i]:
                v_max[i] = cap
        A = np.vstack([D1, np.eye(n)])
        return np.eye(n), -1 * v_max, A, np.hstack([a_min, v_min]), np.hstack([a_max, v_max])

    def compute_speed_profile(self, Constraints, solver=None):
        """Reference velocity per waypoint (src/reference_path.py:289-354).  The QP is solved on the
        device by libmpmpc.so (K4, mpmpc.speed_profile) to a KKT-certified optimum; `solver` is the
        test hook for the CPU emulation of that kernel (same signature).  No host solver behind it."""
        import mpmpc
        n = self.n_waypoints - 1
        li = np.array([self.get_waypoint(i + 1) - self.get_waypoint(i) for i in range(n)], float)
        kappa = np.array([float(self.get_waypoint(i).kappa) for i in range(n)])
        limits = [Constraints[k] for k in ('a_min', 'a_max', 'v_min', 'v_max', 'ay_max')]
        v, status, _ = (solver or mpmpc.speed_profile)(li, kappa, limits, eps=self.eps)
        if int(status[0]) < 0:
            raise ValueError("speed profile: inconsistent constraints")
        v = v[0]
        for i, wp in enumerate(self.waypoints[:-1]):
            wp.v_ref = v[i]
        self.waypoints[-1].v_ref = self.waypoints[-2].v_ref
        self.tables_version = getattr(self, "tables_version", 0) + 1     # the controller re-uploads its path tables

    # ------------------------------------------------------------------ access
    def get_waypoint(self, wp_id):
        if wp_id >= self.n_waypoints:
            if not self.circular:
                print('Reached end of path!')
                raise SystemExit(1)
            wp_id = np.mod(wp_id, self.n_waypoints)
        return self.waypoints[wp_id]

    def tables(self):
        """kappa, v_ref, ds_next per waypoint (float64) for the device path tables."""
        n = self.n_waypoints
        kappa = np.array([float(w.kappa) for w in self.waypoints])
        v_ref = np.array([np.nan if w.v_ref is None else float(w.v_ref) for w in self.waypoints])
        nxt = [(i + 1) % n if self.circular else min(i + 1, n - 1) for i in range(n)]
        ds = np.array([self.waypoints[j] - self.waypoints[i] for i, j in enumerate(nxt)])
        return kappa, v_ref, ds

    # ------------------------------------------------------------------ dynamic corridor
    def _compute_free_segments(self, wp, min_width):
        """Free runs of the rasterised left->right border segment (src/reference_path.py:466-520)."""
        grid = self.map.data
        left = self.map.w2m(wp.static_border_cells[0][0], wp.static_border_cells[0][1])
        right = self.map.w2m(wp.static_border_cells[1][0], wp.static_border_cells[1][1])
        xs, ys, _ = line_aa(left[0], left[1], right[0], right[1])
        segments = []
        start, end = left, left
        in_free = False
        for x, y in zip(xs[1:].tolist(), ys[1:].tolist()):
            free = grid[y, x] == 1
            if free:
                in_free = True
                end = (x, y)
            if (not free or (x, y) == right) and in_free:
                p_start = self.map.m2w(start[0], start[1])
                p_end = self.map.m2w(x, y)
                if np.sqrt((p_start[0] - p_end[0]) ** 2 + (p_start[1] - p_end[1]) ** 2) > min_width:
                    segments.append((p_start, p_end))
                start = (x, y)
                in_free = False
            elif not free and not in_free:
                start = (x, y)
                end = (x, y)
        return segments

    def update_path_constraints(self, wp_id, N, min_width, safety_margin):
        ub_hor, lb_hor, cells_hor, cells_sm_hor = [], [], [], []
        for n in range(N):
            wp = self.get_waypoint(wp_id + n)
            segments = self._compute_free_segments(wp, min_width)
            if n == 0:
                spans = [np.sqrt((s[0][0] - s[1][0]) ** 2 + (s[0][1] - s[1][1]) ** 2) for s in segments]
                pick_u, pick_l = segments[spans.index(max(spans))]
            else:
                prev_u, prev_l = (list(c) for c in cells_hor[n - 1])
                wp_prev = self.get_waypoint(wp_id + n - 1)
                shift = wp_prev - wp
                prev_u[0] += shift * np.cos(wp_prev.psi)
                prev_u[1] += shift * np.cos(wp_prev.psi)
                prev_l[0] += shift * np.sin(wp_prev.psi)
                prev_l[1] += shift * np.sin(wp_prev.psi)
                if len(segments) >= 2:
                    offsets = []
                    for seg_u, seg_l in segments:
                        d_u = np.sqrt((seg_u[0] - prev_u[0]) ** 2 + (seg_u[1] - prev_u[1]) ** 2)
                        d_l = np.sqrt((seg_l[0] - prev_l[0]) ** 2 + (seg_l[1] - prev_l[1]) ** 2)
                        offsets.append((d_u + d_l) / 2)
                    pick_u, pick_l = segments[offsets.index(min(offsets))]
                elif len(segments) == 1:
                    pick_u, pick_l = segments[0]
                else:
                    pick_u, pick_l = (wp.x, wp.y), (wp.x, wp.y)
            side_u = np.sign(_wrap(np.arctan2(pick_u[1] - wp.y, pick_u[0] - wp.x) - wp.psi))
            side_l = np.sign(_wrap(np.arctan2(pick_l[1] - wp.y, pick_l[0] - wp.x) - wp.psi))
            ub = side_u * np.sqrt((pick_u[0] - wp.x) ** 2 + (pick_u[1] - wp.y) ** 2)
            lb = side_l * np.sqrt((pick_l[0] - wp.x) ** 2 + (pick_l[1] - wp.y) ** 2)
            ub -= safety_margin
            lb += safety_margin
            if ub < lb:
                ub, lb = 0.0, 0.0
            ang_u = np.mod(math.pi / 2 + wp.psi + math.pi, 2 * math.pi) - math.pi
            ang_l = np.mod(-math.pi / 2 + wp.psi + math.pi, 2 * math.pi) - math.pi
            cell_u_sm = wp.x + ub * np.cos(ang_u), wp.y + ub * np.sin(ang_u)
            cell_l_sm = wp.x - lb * np.cos(ang_l), wp.y - lb * np.sin(ang_l)
            cell_u = wp.x + (ub + safety_margin) * np.cos(ang_u), wp.y + (ub + safety_margin) * np.sin(ang_u)
            cell_l = wp.x - (lb - safety_margin) * np.cos(ang_l), wp.y - (lb - safety_margin) * np.sin(ang_l)
            ub_hor.append(ub)
            lb_hor.append(lb)
            cells_hor.append([cell_u, cell_l])
            cells_sm_hor.append([cell_u_sm, cell_l_sm])
            wp.dynamic_border_cells = (cell_u_sm, cell_l_sm)
        return np.array(ub_hor), np.array(lb_hor), cells_sm_hor

    def corridor_table(self, n_cols, min_width, safety_margin):
        """update_path_constraints(w + 1, n_cols, ...) for every start waypoint w: for a static map
        the corridor depends on the waypoint index only, so one table serves every later call."""
        ub = np.full((self.n_waypoints, n_cols), np.nan)
        lb = np.full((self.n_waypoints, n_cols), np.nan)
        keep = [w.dynamic_border_cells for w in self.waypoints]
        for w in range(self.n_waypoints):
            try:
                ub[w], lb[w], _ = self.update_path_constraints(w + 1, n_cols, min_width, safety_margin)
            except ValueError:        # no free segment at the first horizon waypoint
                pass
        for wp, cells in zip(self.waypoints, keep):
            wp.dynamic_border_cells = cells
        return ub, lb

    # ------------------------------------------------------------------ drawing
    def show(self, display_drivable_area=True):
        import matplotlib.pyplot as plt
        m = self.map
        plt.clf()
        plt.xticks([])
        plt.yticks([])
        extent = [m.origin[0], m.origin[0] + m.width * m.resolution, m.origin[1], m.origin[1] + m.height * m.resolution]
        plt.imshow(np.ones(m.data.shape), cmap='gray', extent=extent, vmin=0.0, vmax=1.0)
        px = np.array([w.x for w in self.waypoints])
        py = np.array([w.y for w in self.waypoints])
        su = np.array([w.static_border_cells[0] for w in self.waypoints])
        sl = np.array([w.static_border_cells[1] for w in self.waypoints])
        plt.scatter(px, py, c=WAYPOINTS, s=10)
        if display_drivable_area:
            for edge in (su, sl):
                plt.quiver(px, py, edge[:, 0] - px, edge[:, 1] - py, scale=1, units='xy',
                           width=0.2 * self.resolution, color=DRIVABLE_AREA, headwidth=1, headlength=0)
        closed_u, closed_l = np.vstack([su, su[:1]]), np.vstack([sl, sl[:1]])
        if self.circular:
            plt.plot(closed_u[:, 0], closed_u[:, 1], color='#5E5E5E')
            plt.plot(closed_l[:, 0], closed_l[:, 1], color='#5E5E5E')
        else:
            plt.plot(su[:, 0], su[:, 1], color=OBSTACLE)
            plt.plot(sl[:, 0], sl[:, 1], color=OBSTACLE)
            for end in (-1, 0):
                plt.plot((su[end, 0], sl[end, 0]), (su[end, 1], sl[end, 1]), color=OBSTACLE)
        du = np.array([w.dynamic_border_cells[0] for w in self.waypoints] + [self.waypoints[0].static_border_cells[0]])
        dl = np.array([w.dynamic_border_cells[1] for w in self.waypoints] + [self.waypoints[0].static_border_cells[1]])
        plt.plot(du[:, 0], du[:, 1], c=PATH_CONSTRAINTS)
        plt.plot(dl[:, 0], dl[:, 1], c=PATH_CONSTRAINTS)
        for ob in m.obstacles:
            ob.show()
