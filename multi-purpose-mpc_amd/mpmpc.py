"""ctypes binding of the C ABI declared in include/mpmpc.h (libmpmpc.so, HIP kernels for gfx950).

This is the only place Python touches the device library.  There is no CPU fallback: if the
shared library is missing or no HIP device is present, construction of a `Handle` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libmpmpc.so")

NX, NU = 3, 2
NUM_FIELDS = 27
MAX_HORIZON = 255

SOLVED, SOLVED_INACCURATE = 1, 2
MAX_ITER_REACHED, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, UNSOLVED = -2, -3, -4, -10


class Config(C.Structure):
    """mpmpc_config"""
    _fields_ = [("N", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32), ("circular", C.c_int32),
                ("Q", C.c_double * 3), ("R", C.c_double * 2), ("QN", C.c_double * 3),
                ("xmin", C.c_double * 3), ("xmax", C.c_double * 3),
                ("umin", C.c_double * 2), ("umax", C.c_double * 2),
                ("ay_max", C.c_double), ("wheelbase", C.c_double), ("QN_offdiag", C.c_double * 3),
                ("Q_offdiag", C.c_double * 3), ("R_offdiag", C.c_double * 1)]


class Settings(C.Structure):
    """mpmpc_settings (OSQP 0.6.x names/defaults for the ADMM stage + certified polish)"""
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
                ("eps_abs", C.c_double), ("eps_rel", C.c_double),
                ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
                ("max_iter", C.c_int32), ("check_termination", C.c_int32), ("scaling", C.c_int32),
                ("adaptive_rho", C.c_int32), ("adaptive_rho_interval", C.c_int32),
                ("adaptive_rho_tolerance", C.c_double),
                ("polish", C.c_int32), ("ipm_max_iter", C.c_int32),
                ("ipm_tol", C.c_double), ("ipm_reg", C.c_double), ("as_delta", C.c_double),
                ("as_refine", C.c_int32), ("as_rounds", C.c_int32), ("cert_tol", C.c_double),
                ("early_polish", C.c_int32), ("early_scaling", C.c_int32), ("phase1", C.c_int32),
                ("ipm_diverged", C.c_double), ("phase1_theta", C.c_double), ("phase1_eps", C.c_double),
                ("reduce", C.c_int32),
                ("ipm_start_slack", C.c_double), ("ipm_start_mu", C.c_double),
                ("ipm_start_dual", C.c_double), ("as_add_fraction", C.c_double), ("phase1_accept", C.c_int32), ("native", C.c_int32),
                ("native_ipm_tol", C.c_double), ("early_start", C.c_int32), ("phase1_band", C.c_double)]


def default_settings(**kw) -> Settings:
    s = Settings(rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4,
                 eps_dual_inf=1e-4, max_iter=4000, check_termination=25, scaling=10, adaptive_rho=1,
                 adaptive_rho_interval=50, adaptive_rho_tolerance=5.0, polish=2, ipm_max_iter=30,
                 ipm_tol=1e-8, ipm_reg=1e-8, as_delta=1e-10, as_refine=5, as_rounds=4, cert_tol=1e-8, early_polish=1,
                 early_scaling=1, phase1=1, ipm_diverged=1e2, phase1_theta=1.0, phase1_eps=1e-6, reduce=1,
                 ipm_start_slack=0.1, ipm_start_mu=0.01, ipm_start_dual=0.2, as_add_fraction=0.25, phase1_accept=1, native=1, native_ipm_tol=1e-7, early_start=0, phase1_band=3.0)
    for k, v in kw.items():
        if not hasattr(s, k):
            raise TypeError("unknown solver setting %r" % k)
        setattr(s, k, v)
    return s


def stock_settings(**kw) -> Settings:
    """The reference's own solver call (src/MPC.py:158-159,183): OSQP at its defaults and nothing else - no polish, no
    phase 1; statuses and iterates are those of the restated OSQP (eps = 1e-3: up to ~1 rad away from the optimum in
    delta_0 on this problem family)."""
    return default_settings(polish=0, early_polish=0, phase1=0, reduce=0, **kw)


def make_config(N, Q, R, QN, xmin, xmax, umin, umax, ay_max, wheelbase, circular=True, max_batch=1,
                device=0) -> Config:
    if not 3 <= int(N) <= MAX_HORIZON:
        raise ValueError("horizon N must satisfy 3 <= N <= %d" % MAX_HORIZON)
    c = Config(N=int(N), max_batch=int(max_batch), device=int(device), circular=int(bool(circular)),
               ay_max=float(ay_max), wheelbase=float(wheelbase))
    # The reference puts the WHOLE weight matrices into the Hessian (src/MPC.py:150): a matrix argument is split into its
    # diagonal and its off-diagonal entries; a vector argument is a diagonal.
    def split(M, n, name):
        M = np.asarray(M.toarray() if hasattr(M, "toarray") else M, float)
        if M.shape != (n, n):
            return M, None
        if not np.array_equal(M, M.T):
            raise ValueError("%s must be symmetric" % name)
        return np.diag(M).copy(), [M[i, j] for i in range(n) for j in range(i + 1, n)]
    QN, off = split(QN, 3, "QN")
    if off is not None:
        c.QN_offdiag = (C.c_double * 3)(*off)
    Q, off = split(Q, 3, "Q")
    if off is not None:
        c.Q_offdiag = (C.c_double * 3)(*off)
    R, off = split(R, 2, "R")
    if off is not None:
        c.R_offdiag = (C.c_double * 1)(*off)
    for name, val, n in (("Q", Q, 3), ("R", R, 2), ("QN", QN, 3), ("xmin", xmin, 3), ("xmax", xmax, 3),
                         ("umin", umin, 2), ("umax", umax, 2)):
        a = np.asarray(val, float).ravel()
        if a.size != n:
            raise ValueError("%s must have %d entries" % (name, n))
        setattr(c, name, (C.c_double * n)(*a))
    return c


def stage_ld(N: int) -> int:
    for ld in (16, 32, 64, 128):
        if N + 1 <= ld:
            return ld
    return 256


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


_lib = None


def load_library(path: str | None = None):
    """dlopen libmpmpc.so and declare every entry point of include/mpmpc.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("mpmpc: %s not found - build it with `python __graft_entry__.py` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    lib = C.CDLL(p)
    h = C.c_void_p
    lib.mpmpc_version.restype = C.c_char_p
    lib.mpmpc_last_error.restype = C.c_char_p
    lib.mpmpc_device_count.argtypes = [_ip]
    lib.mpmpc_default_settings.argtypes = [C.POINTER(Settings)]
    lib.mpmpc_default_settings.restype = None
    lib.mpmpc_create.argtypes = [C.POINTER(Config), C.POINTER(Settings), C.POINTER(h)]
    lib.mpmpc_destroy.argtypes = [h]
    lib.mpmpc_set_settings.argtypes = [h, C.POINTER(Settings)]
    lib.mpmpc_set_packing.argtypes = [h, C.c_int32]
    lib.mpmpc_set_tail_kernel.argtypes = [h, C.c_int32]
    lib.mpmpc_set_pipeline.argtypes = [h, C.c_int32]
    lib.mpmpc_set_path.argtypes = [h, C.c_int32, _dp, _dp, _dp]
    lib.mpmpc_set_corridor.argtypes = [h, C.c_int32, C.c_int32, _dp, _dp]
    lib.mpmpc_set_map.argtypes = [h, C.c_int32, C.c_int32, C.POINTER(C.c_int8), C.c_double, C.c_double, C.c_double]
    lib.mpmpc_set_path_geometry.argtypes = [h, C.c_int32, _dp, _dp, _dp, _dp, _dp]
    lib.mpmpc_build_corridor.argtypes = [h, C.c_int32, C.c_double, C.c_double, _dp, _dp, _ip]
    lib.mpmpc_rollout_init.argtypes = [h, C.c_int32, C.c_double, _dp, _dp, _dp, _dp]
    lib.mpmpc_rollout_step.argtypes = [h, C.c_int32, C.c_int32]
    lib.mpmpc_rollout_set_counters.argtypes = [h, C.c_int32, _ip]
    lib.mpmpc_rollout_warm_start.argtypes = [h, C.c_int32]
    lib.mpmpc_rollout_state.argtypes = [h, C.c_int32, _dp, _dp, _dp, _ip, _dp, _dp, _ip, _ip, _ip]
    lib.mpmpc_assemble.argtypes = [h, C.c_int32, _ip, _dp, _dp, _dp, _dp, _dp]
    lib.mpmpc_stage_ld.argtypes = [C.c_int32]
    lib.mpmpc_stage_ld.restype = C.c_int32
    lib.mpmpc_solve.argtypes = [h, C.c_int32, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp]
    lib.mpmpc_upload.argtypes = [h, C.c_int32, _ip, _dp, _dp, _dp, _dp]
    lib.mpmpc_solve_resident.argtypes = [h, C.c_int32]
    lib.mpmpc_set_outputs.argtypes = [h, C.c_int32]
    lib.mpmpc_sync.argtypes = [h]
    lib.mpmpc_download.argtypes = [h, C.c_int32, _dp, _dp, _ip, _ip, _dp, _dp]
    lib.mpmpc_solve_resident_timed.argtypes = [h, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.mpmpc_solve_resident_profile.argtypes = [h, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.mpmpc_assemble_resident_timed.argtypes = [h, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    lib.mpmpc_speed_profile.argtypes = [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, C.c_double, _dp, _ip, _ip]
    _ipp, _dpp = C.POINTER(_ip), C.POINTER(_dp)
    lib.mpmpc_staging.argtypes = [h, C.c_int32, _ipp, _dpp, _dpp, _dpp, _dpp, _dpp, _dpp, _ipp, _ipp, _dpp, _dpp]
    lib.mpmpc_solve_staged.argtypes = [h, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.mpmpc_staged_begin.argtypes = [h, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.mpmpc_staged_end.argtypes = [h]
    if path is None:
        _lib = lib
    return lib


EXPORTS = ["mpmpc_version", "mpmpc_last_error", "mpmpc_device_count", "mpmpc_default_settings",
           "mpmpc_create", "mpmpc_destroy", "mpmpc_set_settings", "mpmpc_set_packing", "mpmpc_set_tail_kernel", "mpmpc_set_path", "mpmpc_set_corridor",
           "mpmpc_set_map", "mpmpc_set_path_geometry", "mpmpc_build_corridor", "mpmpc_rollout_init",
           "mpmpc_rollout_step", "mpmpc_rollout_set_counters", "mpmpc_rollout_warm_start", "mpmpc_rollout_state", "mpmpc_assemble", "mpmpc_stage_ld", "mpmpc_solve", "mpmpc_upload", "mpmpc_solve_resident", "mpmpc_set_outputs", "mpmpc_set_pipeline",
           "mpmpc_sync", "mpmpc_download", "mpmpc_solve_resident_timed", "mpmpc_solve_resident_profile", "mpmpc_assemble_resident_timed", "mpmpc_speed_profile", "mpmpc_staging",
           "mpmpc_solve_staged", "mpmpc_staged_begin", "mpmpc_staged_end"]


class MpmpcError(RuntimeError):
    pass


class Solution:
    """Outputs of one batched solve (host arrays)."""
    __slots__ = ("z", "u0", "status", "iters", "resid", "y")

    def __init__(self, z, u0, status, iters, resid, y):
        self.z, self.u0, self.status, self.iters, self.resid, self.y = z, u0, status, iters, resid, y


class _StagingView(np.ndarray):
    """ndarray view into a handle's staging block; carries a reference to the Handle that owns the memory"""
    _owner = None

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


class Handle:
    """One device context: controller constants, path tables, device buffers for max_batch QPs."""

    def __init__(self, config: Config, settings: Settings | None = None):
        self.lib = load_library()
        self.cfg = config
        self.N = config.N
        self.n = 5 * self.N + 3
        self.m = 8 * self.N + 6
        self.settings = settings or default_settings()
        self._h = C.c_void_p()
        self._check(self.lib.mpmpc_create(C.byref(self.cfg), C.byref(self.settings), C.byref(self._h)))
        self._have_table = False

    def _check(self, rc):
        if rc != 0:
            raise MpmpcError("mpmpc error %d: %s" % (rc, self.lib.mpmpc_last_error().decode()))

    def close(self):
        if self._h:
            self.lib.mpmpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_settings(self, settings: Settings):
        self.settings = settings
        self._check(self.lib.mpmpc_set_settings(self._h, C.byref(settings)))

    def set_packing(self, lanes_per_instance: int = 0):
        """0 = automatic; 64 / 32 / 16 force that many lanes of a wavefront per instance (tests, tuning).  16 at horizons
        16 .. 31: two stages per lane, four instances per wavefront (measured slower than 32: never automatic).  Horizons above
        63: 128 (N <= 127) / 256 (N >= 128) select the one-stage workgroup kernels instead of the two-stages-per-lane default."""
        self._check(self.lib.mpmpc_set_packing(self._h, int(lanes_per_instance)))

    def set_tail_kernel(self, reduced_native: bool = True):
        """True / 1 (default): the reduced-native tail kernel takes the tail of a batch launch first, two instances per wavefront;
        2: the same, one instance per wavefront; False / 0: the general kernel takes all of it (parity tests, A/B timings)."""
        self._check(self.lib.mpmpc_set_tail_kernel(self._h, int(reduced_native)))

    def set_path(self, kappa, v_ref, ds_next):
        k, v, d = (np.ascontiguousarray(a, dtype=np.float64) for a in (kappa, v_ref, ds_next))
        if not (k.size == v.size == d.size):
            raise ValueError("path tables must have equal length")
        self._check(self.lib.mpmpc_set_path(self._h, k.size, _d(k), _d(v), _d(d)))

    def set_corridor(self, ub, lb):
        ub = np.ascontiguousarray(ub, dtype=np.float64)
        lb = np.ascontiguousarray(lb, dtype=np.float64)
        if ub.shape != lb.shape or ub.ndim != 2:
            raise ValueError("corridor tables must be equal-shape [n_wp x n_cols]")
        self._check(self.lib.mpmpc_set_corridor(self._h, ub.shape[0], ub.shape[1], _d(ub), _d(lb)))
        self._have_table = True

    # --- corridor generation on the device (dynamic maps)
    def set_map(self, data, origin, resolution):
        grid = np.ascontiguousarray(data, dtype=np.int8)
        if grid.ndim != 2:
            raise ValueError("map grid must be 2-D [height, width]")
        self._check(self.lib.mpmpc_set_map(self._h, grid.shape[0], grid.shape[1],
                                           grid.ctypes.data_as(C.POINTER(C.c_int8)), float(origin[0]),
                                           float(origin[1]), float(resolution)))

    def set_path_geometry(self, x, y, psi, border_ub, border_lb):
        x, y, psi = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, psi))
        bu = np.ascontiguousarray(border_ub, dtype=np.float64).reshape(x.size, 2)
        bl = np.ascontiguousarray(border_lb, dtype=np.float64).reshape(x.size, 2)
        self._check(self.lib.mpmpc_set_path_geometry(self._h, x.size, _d(x), _d(y), _d(psi), _d(bu), _d(bl)))
        self._n_wp = x.size

    def build_corridor(self, n_cols, min_width, safety_margin, want_tables=True):
        """update_path_constraints(w + 1, n_cols, ...) for every start waypoint w, on the device; the
        handle then uses the result like a table given to set_corridor.  -> (ub, lb, bad_rows)"""
        n = self._n_wp
        ub = np.zeros((n, n_cols)) if want_tables else None
        lb = np.zeros((n, n_cols)) if want_tables else None
        bad = C.c_int32(0)
        self._check(self.lib.mpmpc_build_corridor(self._h, int(n_cols), float(min_width), float(safety_margin),
                                                  _d(ub), _d(lb), C.byref(bad)))
        self._have_table = True
        return ub, lb, bad.value

    # --- closed-loop rollout on the device
    def rollout_init(self, Ts, cum_lengths, s, poses, cc0=None):
        cum = np.ascontiguousarray(cum_lengths, dtype=np.float64)
        s = np.ascontiguousarray(s, dtype=np.float64).ravel()
        B = s.size
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(B, 3)
        if cc0 is not None:
            cc0 = np.ascontiguousarray(cc0, dtype=np.float64).reshape(B, 2 * self.N)
        self._check(self.lib.mpmpc_rollout_init(self._h, B, float(Ts), _d(cum), _d(s), _d(poses), _d(cc0)))
        self._ro_B = B
        return B

    def rollout_set_counters(self, counters):
        """MPC.infeasibility_counter of every car (to resume a recorded run)"""
        c = np.ascontiguousarray(counters, dtype=np.int32).ravel()
        self._check(self.lib.mpmpc_rollout_set_counters(self._h, c.size, _i(c)))

    def rollout_warm_start(self, enable=True):
        """closed loop: try the previous step's shifted active set first.  True / False force it on / off, "auto"
        (the handle's default) uses it for fleets of more than 1024 or at most 16 cars, where it pays."""
        self._check(self.lib.mpmpc_rollout_warm_start(self._h, 2 if enable == "auto" else int(bool(enable))))

    def rollout_step(self, n_steps=1):
        self._check(self.lib.mpmpc_rollout_step(self._h, self._ro_B, int(n_steps)))

    def rollout_state(self):
        B, N = self._ro_B, self.N
        out = dict(s=np.zeros(B), pose=np.zeros((B, 3)), cc=np.zeros((B, 2 * N)), wp_id=np.zeros(B, np.int32),
                   x0=np.zeros((B, 3)), u=np.zeros((B, 2)), status=np.zeros(B, np.int32),
                   counter=np.zeros(B, np.int32), alive=np.zeros(B, np.int32))
        self._check(self.lib.mpmpc_rollout_state(self._h, B, _d(out["s"]), _d(out["pose"]), _d(out["cc"]),
                                                 _i(out["wp_id"]), _d(out["x0"]), _d(out["u"]), _i(out["status"]),
                                                 _i(out["counter"]), _i(out["alive"])))
        return out

    def _inputs(self, wp_id, x0, cc_prev, lb, ub):
        wp = np.ascontiguousarray(wp_id, dtype=np.int32).ravel()
        B = wp.size
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, 3)
        cc = np.ascontiguousarray(cc_prev, dtype=np.float64).reshape(B, 2 * self.N)
        if (lb is None) != (ub is None):
            raise ValueError("lb and ub must both be given or both be None")
        if lb is not None:
            lb = np.ascontiguousarray(lb, dtype=np.float64).reshape(B, self.N)
            ub = np.ascontiguousarray(ub, dtype=np.float64).reshape(B, self.N)
        elif not self._have_table:
            raise ValueError("no per-instance corridor given and no corridor table set")
        return B, wp, x0, cc, lb, ub

    def assemble(self, wp_id, x0, cc_prev, lb=None, ub=None):
        """Stage-blocked QP [NUM_FIELDS, B, LD] as K1 leaves it in HBM."""
        B, wp, x0, cc, lb, ub = self._inputs(wp_id, x0, cc_prev, lb, ub)
        qp = np.zeros((NUM_FIELDS, B, stage_ld(self.N)))
        self._check(self.lib.mpmpc_assemble(self._h, B, _i(wp), _d(x0), _d(cc), _d(lb), _d(ub), _d(qp)))
        return qp

    def _outputs(self, B, want_y):
        return (np.zeros((B, self.n)), np.zeros((B, 2)), np.zeros(B, np.int32), np.zeros((B, 2), np.int32),
                np.zeros((B, 2)), np.zeros((B, self.m)) if want_y else None)

    def solve(self, wp_id, x0, cc_prev, lb=None, ub=None, want_y=False) -> Solution:
        B, wp, x0, cc, lb, ub = self._inputs(wp_id, x0, cc_prev, lb, ub)
        z, u0, st, it, rs, y = self._outputs(B, want_y)
        self._check(self.lib.mpmpc_solve(self._h, B, _i(wp), _d(x0), _d(cc), _d(lb), _d(ub), _d(z), _d(u0),
                                         _i(st), _i(it), _d(rs), _d(y)))
        return Solution(z, u0, st, it, rs, y)

    # --- zero-copy host form: numpy views of the handle's page-locked staging blocks
    def staging(self, B):
        """-> dict of numpy views (wp_id [B], x0 [B,3], cc_prev [B,2N], lb / ub [B,N]; z [B,n], u0 [B,2], status [B], iters [B,2],
        resid [B,2], y [B,m]) laid out for a batch of B: fill the inputs in place, call solve_staged(B), read the outputs."""
        N = self.cfg.N
        pi, pd = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        wp, st, it = pi(), pi(), pi()
        x0, cc, lb, ub, z, u0, rs, y = (pd() for _ in range(8))
        self._check(self.lib.mpmpc_staging(self._h, B, C.byref(wp), C.byref(x0), C.byref(cc), C.byref(lb), C.byref(ub), C.byref(z),
                                           C.byref(u0), C.byref(st), C.byref(it), C.byref(rs), C.byref(y)))
        def view(p, shape):
            # (the views point into the handle's page-locked blocks: they keep the Handle alive - its close() / __del__ frees them)
            v = np.ctypeslib.as_array(p, shape=shape).view(_StagingView)
            v._owner = self
            return v
        return dict(wp_id=view(wp, (B,)), x0=view(x0, (B, 3)), cc_prev=view(cc, (B, 2 * N)), lb=view(lb, (B, N)), ub=view(ub, (B, N)),
                    z=view(z, (B, self.n)), u0=view(u0, (B, 2)), status=view(st, (B,)), iters=view(it, (B, 2)), resid=view(rs, (B, 2)),
                    y=view(y, (B, self.m)))

    def solve_staged(self, B, with_rows=True, want_z=True, want_y=False):
        self._check(self.lib.mpmpc_solve_staged(self._h, B, int(with_rows), int(want_z), int(want_y)))

    def staged_begin(self, B, with_rows=True, want_z=True, want_y=False):
        """solve_staged in two halves (a loop that keeps several handles busy): enqueue, return; staged_end() waits."""
        self._check(self.lib.mpmpc_staged_begin(self._h, B, int(with_rows), int(want_z), int(want_y)))

    def staged_end(self):
        self._check(self.lib.mpmpc_staged_end(self._h))

    # --- resident (benchmark / closed loop) form
    def upload(self, wp_id, x0, cc_prev, lb=None, ub=None):
        B, wp, x0, cc, lb, ub = self._inputs(wp_id, x0, cc_prev, lb, ub)
        self._check(self.lib.mpmpc_upload(self._h, B, _i(wp), _d(x0), _d(cc), _d(lb), _d(ub)))
        return B

    def solve_resident(self, B):
        self._check(self.lib.mpmpc_solve_resident(self._h, B))

    def set_outputs(self, want_y=True):
        """resident launches store the multipliers y (default) or skip them (46 % of the output bytes)"""
        self._check(self.lib.mpmpc_set_outputs(self._h, int(bool(want_y))))

    def set_pipeline(self, depth=3):
        """resident launches in flight, 1 .. 8: 3 (default) = pipelined inside the handle (launch k + 1, k + 2 run beside
        launch k, whose results stay readable until launch k + 3), 1 = one stream, one output block; more than 3 pay only with
        GPU_MAX_HW_QUEUES >= 8 in the environment before the first HIP call (streams sharing a hardware queue serialise)"""
        self._check(self.lib.mpmpc_set_pipeline(self._h, int(depth)))

    def sync(self):
        self._check(self.lib.mpmpc_sync(self._h))

    def download(self, B, want_y=False) -> Solution:
        z, u0, st, it, rs, y = self._outputs(B, want_y)
        self._check(self.lib.mpmpc_download(self._h, B, _d(z), _d(u0), _i(st), _i(it), _d(rs), _d(y)))
        return Solution(z, u0, st, it, rs, y)

    def solve_resident_timed(self, B):
        a, s = C.c_float(), C.c_float()
        self._check(self.lib.mpmpc_solve_resident_timed(self._h, B, C.byref(a), C.byref(s)))
        return a.value, s.value

    def solve_resident_profile(self, B, n):
        """n resident launches as solve_resident issues them (double-buffered), HIP events around each on its own stream
        -> (durations [n] in ms, span from the first start to the last end in ms)"""
        each = np.zeros(n, np.float32)
        span = C.c_float()
        self._check(self.lib.mpmpc_solve_resident_profile(self._h, B, n, each.ctypes.data_as(C.POINTER(C.c_float)), C.byref(span)))
        return each.astype(float), span.value

    def assemble_timed(self, B, n):
        """n launches of the stand-alone assembly kernel K1 back to back, HIP events around each -> durations [n] in ms"""
        each = np.zeros(n, np.float32)
        self._check(self.lib.mpmpc_assemble_resident_timed(self._h, B, n, each.ctypes.data_as(C.POINTER(C.c_float))))
        return each.astype(float)


def device_count() -> int:
    n = C.c_int32(0)
    rc = load_library().mpmpc_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def speed_profile(li, kappa, limits, eps=1e-12, device=0):
    """Speed profiles of B paths on the device (K4, replaces compute_speed_profile's OSQP call,
    src/reference_path.py:289-354).  li, kappa [B, n] (or [n]); limits [B, 5] (or [5]) =
    (a_min, a_max, v_min, v_max, ay_max).  -> (v [B, n], status [B], iters [B])."""
    lib = load_library()
    li = np.atleast_2d(np.ascontiguousarray(li, float))
    kappa = np.atleast_2d(np.ascontiguousarray(kappa, float))
    B, n = li.shape
    if kappa.shape != (B, n):
        raise ValueError("li and kappa must have the same shape")
    limits = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(limits, float)), (B, 5)))
    v = np.zeros((B, n))
    status, iters = np.zeros(B, np.int32), np.zeros(B, np.int32)
    rc = lib.mpmpc_speed_profile(device, B, n, _d(li), _d(kappa), _d(limits), float(eps), _d(v), _i(status), _i(iters))
    if rc != 0:
        raise MpmpcError("mpmpc_speed_profile: %s (rc=%d)" % (lib.mpmpc_last_error().decode(), rc))
    return v, status, iters
