"""Model predictive controller on the GPU QP path.

`MPC` keeps the constructor, attributes and `get_control()` contract of the reference's
src/MPC.py:14-257 so that src/simulation.py's body runs against it unchanged:

    mpc = MPC(car, N, Q, R, QN, StateConstraints, InputConstraints, ay_max)
    u = mpc.get_control()            # np.array([v, delta]); u[0], u[1]
    mpc.show_prediction()

What changed is where the work happens.  The reference rebuilds dense A/B, scipy.sparse P/A and a
fresh OSQP workspace every step (src/MPC.py:61-159) and solves on one CPU thread
(src/MPC.py:183).  Here the per-waypoint tables are uploaded once, and every step sends
(waypoint id, spatial state, previous plan, corridor) through the C ABI of libmpmpc.so, where one
HIP kernel assembles the stage-blocked QP and a second one solves it (OSQP-style ADMM + certified
polish).  There is no CPU solver behind this class: without the library / a device it raises.

`BatchMPC` is the one API extension: B independent controller instances per call (scenario
sweeps, Monte-Carlo initial poses) on one device.

Weights: Q, R, QN may be any symmetric positive semidefinite matrices (scipy sparse or dense).  Like
the reference, the Hessian takes the whole matrices (src/MPC.py:150) while the cost vector uses
diag(Q), diag(R) and the whole QN (src/MPC.py:153-155).  Diagonal weights (src/simulation.py:101-103)
run the reduced-native kernels; anything else the general kernels.  3 <= N <= mpmpc.MAX_HORIZON.
"""
from __future__ import annotations

import numpy as np

import mpmpc
from spatial_bicycle_models import current_waypoint_batch, t2s_batch

PREDICTION = '#BA4A00'


def _symmetric(M, n, name):
    """a weight matrix (scipy sparse or dense) as a dense symmetric positive semidefinite array - anything else is refused"""
    D = M.toarray() if hasattr(M, "toarray") else np.asarray(M, float)
    D = np.atleast_2d(D).astype(float)
    if D.shape != (n, n):
        raise ValueError("%s must be %dx%d" % (name, n, n))
    if not np.array_equal(D, D.T):
        raise ValueError("%s must be symmetric" % name)
    if not np.all(np.isfinite(D)) or np.linalg.eigvalsh(D).min() < -1e-12 * (1.0 + np.abs(D).max()):
        raise ValueError("%s must be finite and positive semidefinite" % name)
    return D


def _make_config(model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, max_batch, device):
    rp = model.reference_path
    return mpmpc.make_config(N, _symmetric(Q, 3, "Q"), _symmetric(R, 2, "R"), _symmetric(QN, 3, "QN"),
                             StateConstraints['xmin'], StateConstraints['xmax'],
                             InputConstraints['umin'], InputConstraints['umax'], ay_max, model.length,
                             circular=rp.circular, max_batch=max_batch, device=device)


class MPC:
    def __init__(self, model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max,
                 settings=None, device=0, backend=None, corridor="host"):
        """`corridor`: "host" calls ReferencePath.update_path_constraints every step like the reference
        (src/MPC.py:116-118; 5 ms of Python here, 17 ms there, and it refreshes the plotted border cells);
        "device" keeps the table of all start waypoints on the GPU (K0, rebuilt when the map's grid changes)
        and the step sends no bounds at all - the whole get_control() is then ~0.3 ms."""
        if corridor not in ("host", "device"):
            raise ValueError("corridor must be 'host' or 'device'")
        self.corridor = corridor
        self._map_version = None
        self.N = N
        self.Q, self.R, self.QN = Q, R, QN
        self.model = model
        self.nx = self.model.n_states
        self.nu = 2
        self.state_constraints = StateConstraints
        self.input_constraints = InputConstraints
        self.ay_max = ay_max
        self.current_prediction = None
        self.infeasibility_counter = 0
        self.current_control = np.zeros(self.nu * self.N)
        self.settings = settings or mpmpc.default_settings()
        self._cfg = _make_config(model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, 1, device)
        # `backend`: anything with set_path / solve like mpmpc.Handle (the tests inject the CPU
        # emulation of the kernels); the product default is the HIP library and nothing else
        self.optimizer = backend if backend is not None else mpmpc.Handle(self._cfg, self.settings)
        self._path_version = None
        self._path_stamp = None
        self.last_status = None
        self.last_solution = None

    # -- path tables follow the ReferencePath (v_ref is filled by compute_speed_profile after
    #    the controller is constructed, src/simulation.py:112-119)
    def _sync_path(self):
        rp = self.model.reference_path
        # cheap check first: compute_speed_profile bumps the version; a path without one (v_ref assigned by hand,
        # as some tests do) is compared by content
        stamp = (id(rp), getattr(rp, "tables_version", None))
        if stamp[1] is not None and stamp == self._path_stamp:
            return
        kappa, v_ref, ds = rp.tables()
        key = (kappa.tobytes(), v_ref.tobytes(), ds.tobytes())
        if key != self._path_version:
            if np.any(np.isnan(v_ref)):
                raise RuntimeError("reference path has no speed profile (call compute_speed_profile first)")
            self.optimizer.set_path(kappa, v_ref, ds)
            self._path_version = key
        self._path_stamp = stamp

    def _init_problem(self):
        """Inputs of this step's QP (the arithmetic of src/MPC.py:61-155 happens in K1 on the device)."""
        rp, m = self.model.reference_path, self.model
        self._sync_path()
        x0 = np.array(m.spatial_state[:], dtype=float)
        head = (np.array([m.wp_id], dtype=np.int32), x0[None, :], np.asarray(self.current_control, float)[None, :])
        if self.corridor == "device":
            self._sync_corridor()
            return head + (None, None)
        ub, lb, _ = rp.update_path_constraints(m.wp_id + 1, self.N, 2 * m.safety_margin, m.safety_margin)
        return head + (np.asarray(lb, float)[None, :], np.asarray(ub, float)[None, :])

    def _sync_corridor(self):
        """Device corridor table (row w = update_path_constraints(w + 1, N, 2 sm, sm)) follows the map: it is
        rebuilt (two small launches) whenever the occupancy grid differs from the one it was built from."""
        rp, grid = self.model.reference_path, self.model.reference_path.map
        version = grid.data.tobytes()
        if version == self._map_version:
            return
        if not hasattr(self.optimizer, "build_corridor"):
            raise RuntimeError("corridor='device' needs the HIP library (mpmpc.Handle)")
        wps = rp.waypoints
        self.optimizer.set_map(grid.data, grid.origin, grid.resolution)
        self.optimizer.set_path_geometry([w.x for w in wps], [w.y for w in wps], [w.psi for w in wps],
                                         [w.static_border_cells[0] for w in wps], [w.static_border_cells[1] for w in wps])
        sm = self.model.safety_margin
        self.optimizer.build_corridor(self.N, 2 * sm, sm, want_tables=False)
        self._map_version = version

    def get_control(self):
        nx, nu = self.model.n_states, 2
        self.model.get_current_waypoint()
        self.model.spatial_state = self.model.t2s(reference_state=self.model.temporal_state,
                                                  reference_waypoint=self.model.current_waypoint)
        wp, x0, cc, lb, ub = self._init_problem()
        sol = self.optimizer.solve(wp, x0, cc, lb, ub)
        self.last_solution = sol
        self.last_status = int(sol.status[0])
        # stock OSQP hands back a usable x for "solved", "solved inaccurate" and "max iter reached";
        # only an infeasibility verdict makes the reference take its fallback branch (src/MPC.py:208)
        if self.last_status in (mpmpc.SOLVED, mpmpc.SOLVED_INACCURATE, mpmpc.MAX_ITER_REACHED):
            z = sol.z[0]
            plan = np.array(z[-self.N * nu:])
            plan[1::2] = np.arctan(plan[1::2] * self.model.length)
            self.current_control = plan
            self.current_prediction = self.update_prediction(np.reshape(z[:(self.N + 1) * nx], (self.N + 1, nx)))
            u = np.array([plan[0], plan[1]])
            self.infeasibility_counter = 0
        else:
            print('Infeasible problem. Previously predicted control signal used!')
            i = nu * (self.infeasibility_counter + 1)
            u = np.array(self.current_control[i:i + 2])
            self.infeasibility_counter += 1
        if self.infeasibility_counter == (self.N - 1):
            print('No control signal computed!')
            raise SystemExit(1)
        return u

    def update_prediction(self, spatial_state_prediction):
        """Predicted e_y per stage -> world x, y for plotting (stages 2..N-1, src/MPC.py:224-248)."""
        rp, wp0 = self.model.reference_path, self.model.wp_id
        xs, ys = [], []
        for n in range(2, self.N):
            w = rp.get_waypoint(wp0 + n)
            p = self.model.s2t(w, spatial_state_prediction[n, :])
            xs.append(p.x)
            ys.append(p.y)
        return xs, ys

    def show_prediction(self):
        if self.current_prediction is not None:
            import matplotlib.pyplot as plt
            plt.scatter(self.current_prediction[0], self.current_prediction[1], c=PREDICTION, s=30)


class BatchMPC:
    """B independent controllers sharing one path, weights and limits; one launch per call."""

    def __init__(self, model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, max_batch,
                 settings=None, device=0, corridor=None):
        self.N, self.model = N, model
        self.settings = settings or mpmpc.default_settings()
        self._cfg = _make_config(model, N, Q, R, QN, StateConstraints, InputConstraints, ay_max, max_batch, device)
        self.handle = mpmpc.Handle(self._cfg, self.settings)
        kappa, v_ref, ds = model.reference_path.tables()
        if np.any(np.isnan(v_ref)):
            raise RuntimeError("reference path has no speed profile (call compute_speed_profile first)")
        self.handle.set_path(kappa, v_ref, ds)
        self._path = model.reference_path
        self.corridor_cols = None
        self._corridor_tables = None      # host copies of a static corridor table (for the further handles of get_control_stream)
        self._stream = None
        self._ring = []                   # the further handles of get_control_stream (built like self.handle, kept in step with it)
        self._packing, self._tail_kernel = None, None
        if isinstance(corridor, str) and corridor == "device":
            self.update_corridor_from_map()
        elif corridor is not None:        # (ub, lb) tables [n_wp x >=N] of a static map
            self.handle.set_corridor(*corridor)
            self._corridor_tables = corridor

    def _handles(self):
        return [self.handle] + list(self._ring)

    def _corridor_from_map(self, h, n_cols):
        rp, m = self._path, self._path.map
        wps = rp.waypoints
        h.set_map(m.data, m.origin, m.resolution)
        h.set_path_geometry([w.x for w in wps], [w.y for w in wps], [w.psi for w in wps],
                            [w.static_border_cells[0] for w in wps], [w.static_border_cells[1] for w in wps])
        sm = self.model.safety_margin
        return h.build_corridor(n_cols, 2 * sm, sm, want_tables=False)[2]

    # Everything that changes what a handle solves goes through these, so that the ring of get_control_stream never works on a
    # stale corridor or stale settings (a call on `self.handle` alone reaches only the first handle of the ring).
    def _quiesce_stream(self):
        """nothing of a stream may still be in flight on a handle whose settings, packing, tail kernel or corridor change: the
        C side would settle the staged call (MPMPC_SETTLE) and a live get_control_stream generator would then yield fewer or
        stale batches - what is in flight is waited for and dropped, the generator ends at its next step"""
        if self._stream is not None:
            self._stream.discard()

    def set_settings(self, settings):
        self._quiesce_stream()
        self.settings = settings
        for h in self._handles():
            h.set_settings(settings)

    def set_packing(self, lanes_per_instance=0):
        self._quiesce_stream()
        self._packing = lanes_per_instance
        for h in self._handles():
            h.set_packing(lanes_per_instance)

    def set_tail_kernel(self, reduced_native=True):
        self._quiesce_stream()
        self._tail_kernel = reduced_native
        for h in self._handles():
            h.set_tail_kernel(reduced_native)

    def set_corridor(self, ub, lb):
        """a static corridor table [n_wp x >= N] for every handle of this controller"""
        self._quiesce_stream()
        for h in self._handles():
            h.set_corridor(ub, lb)
        self._corridor_tables, self.corridor_cols = (ub, lb), None

    def _close_ring(self):
        if self._stream is not None:
            self._stream.discard()
        for h in self._ring:
            h.close()
        self._ring, self._stream = [], None

    def close(self):
        self._close_ring()
        self.handle.close()

    def update_corridor_from_map(self, n_cols=None):
        """(Re)build the corridor table on the device from the path's map as it is NOW (obstacles
        added since the last call included): update_path_constraints(w + 1, n_cols, 2*sm, sm) for every
        start waypoint w (src/MPC.py:116-118, src/reference_path.py:522-648), without leaving the GPU.
        Returns the number of start waypoints whose first horizon waypoint is fully blocked.  Every handle of this
        controller - the ring of get_control_stream included - gets the new table."""
        n_cols = int(n_cols or self.N)
        if self._stream is not None:
            self._stream.discard()        # nothing of a stream may still be in flight on a handle whose table changes
        bad = [self._corridor_from_map(h, n_cols) for h in self._handles()][0]
        self.corridor_cols = n_cols
        self._corridor_tables = None
        return bad

    def spatial_states(self, s, poses):
        """(wp_id[B], x0[B,3]) from arc lengths and world poses, as get_control does per car."""
        rp = self._path
        wp = current_waypoint_batch(rp.segment_lengths, s)
        wx = np.array([w.x for w in rp.waypoints])[wp]
        wy = np.array([w.y for w in rp.waypoints])[wp]
        wpsi = np.array([w.psi for w in rp.waypoints])[wp]
        poses = np.asarray(poses, float)
        return wp.astype(np.int32), t2s_batch(poses[:, 0], poses[:, 1], poses[:, 2], wx, wy, wpsi)

    def rollout(self, s, poses, n_steps, cc0=None):
        """Drive B cars `n_steps` control steps on the device (localise, assemble, solve, fallback,
        plant update: the loop of src/simulation.py:134-140) and return the final state dict
        (s, pose, cc, wp_id, x0, u, status, counter, alive).  Needs a corridor table."""
        rp = self._path
        if getattr(self.handle, "_n_wp", None) != rp.n_waypoints:
            wps = rp.waypoints
            self.handle.set_path_geometry([w.x for w in wps], [w.y for w in wps], [w.psi for w in wps],
                                          [w.static_border_cells[0] for w in wps],
                                          [w.static_border_cells[1] for w in wps])
        self.handle.rollout_init(self.model.Ts, np.cumsum(rp.segment_lengths), s, poses, cc0)
        self.handle.rollout_step(n_steps)
        return self.handle.rollout_state()

    def staging(self, B):
        """numpy views of the handle's page-locked staging blocks for a batch of B (mpmpc.Handle.staging): a caller that builds
        wp_id / x0 / cc_prev (/ lb / ub) in place and reads u0 / status / z in place skips the host-side copies, which are two
        thirds of get_control_batch at B = 1 024."""
        return self.handle.staging(B)

    def get_control_staged(self, B, with_rows=True, want_plan=True):
        """get_control_batch on the staging views: -> (u [B,2] = (v, delta), status [B]) as views into the staging block; the
        plan is in staging(B)["z"] (kappa entries, not yet delta) when want_plan."""
        self.handle.solve_staged(B, with_rows=with_rows, want_z=want_plan, want_y=False)
        st = self.handle.staging(B)
        return st["u0"], st["status"]

    def get_control_stream(self, batches, depth=3, want_plan=True):
        """A stream of get_control_batch calls from host buffers with `depth` of them in flight on this device
        (streamed.StreamedBatches: upload, launch and download of consecutive batches overlap).  batches: iterable of
        (wp_id, x0, cc_prev[, lb, ub]); yields (u [B,2] = (v, delta), plan [B,2N] with delta entries or None, status [B]) per
        batch, in order.  The corridor comes with each batch (lb / ub) or from the table this controller was given or built
        (corridor=(ub, lb), corridor="device" / update_corridor_from_map: every handle of the ring holds it).  Settings,
        packing and tail kernel follow this controller's set_settings / set_packing / set_tail_kernel."""
        import streamed
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if self._stream is None or self._stream.depth != depth:
            # (the first handle of the ring is this controller's own; the others are built like it and carry its corridor,
            #  settings, packing and tail kernel; a ring of another depth is closed first: each handle holds output blocks and
            #  pinned staging)
            self._close_ring()
            for _ in range(depth - 1):
                h = mpmpc.Handle(self._cfg, self.settings)
                h.set_path(*self._path.tables())
                if self._corridor_tables is not None:
                    h.set_corridor(*self._corridor_tables)
                elif self.corridor_cols is not None:
                    self._corridor_from_map(h, self.corridor_cols)
                if self._packing is not None:
                    h.set_packing(self._packing)
                if self._tail_kernel is not None:
                    h.set_tail_kernel(self._tail_kernel)
                self._ring.append(h)
            self._stream = streamed.StreamedBatches(handles=self._handles())
        for sol in self._stream.map(batches, want_z=want_plan):
            plan = None
            if want_plan:
                plan = sol.z[:, -2 * self.N:].copy()
                plan[:, 1::2] = np.arctan(plan[:, 1::2] * self.model.length)
            yield sol.u0, plan, sol.status

    def get_control_batch(self, wp_id, x0, cc_prev, lb=None, ub=None):
        """-> (u [B,2] = (v, delta), plan [B,2N] with delta entries, status [B], Solution)."""
        sol = self.handle.solve(wp_id, x0, cc_prev, lb, ub)
        plan = sol.z[:, -2 * self.N:].copy()
        plan[:, 1::2] = np.arctan(plan[:, 1::2] * self.model.length)
        return sol.u0, plan, sol.status, sol
