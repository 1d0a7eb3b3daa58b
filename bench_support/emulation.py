"""The kernels' CPU lock-step emulation behind Python objects: `Emul` (ctypes view of tests/_build/libmpmpc_emul*.so - the lane code
of multi-purpose-mpc_amd/csrc compiled for the host, built by tests/emul/Makefile), `EmuBackend` (drop-in for mpmpc.Handle in
host-logic tests) and `DryHandle` (the handle's RESIDENT surface, what `bench.py --dry-run` drives when it rehearses the
multi-rank plumbing on a box without GPUs).  TEST / REHEARSAL INFRASTRUCTURE: never imported by the product package, and
nothing here is ever measured - the emulation is a checker.  Lives outside tests/ because bench.py uses it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multi-purpose-mpc_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import mpmpc  # noqa: E402

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


class Emul:
    """ctypes view of tests/_build/libmpmpc_emul.so (CPU lock-step emulation of the kernels)."""

    def __init__(self):
        so = os.path.join(ROOT, "tests", "_build", "libmpmpc_emul.so")
        subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "tests", "emul")], check=True)
        self.lib = C.CDLL(so)
        self._wide = {}

    def wide(self, width):
        """the general solver on an emulated WORKGROUP of 128 / 256 lanes (horizons 64 .. 255: tests/emul/emul_wide.cpp)"""
        if width not in self._wide:
            self._wide[width] = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmpmpc_emul_w%d.so" % width))
            assert self._wide[width].emuw_width() == width
        return self._wide[width]

    def assemble(self, cfg, track, inputs, use_table=False, obstacles=False):
        wp, x0, cc, lb, ub = inputs
        B = wp.size
        N = cfg.N
        ld = mpmpc.stage_ld(N)
        qp = np.zeros((mpmpc.NUM_FIELDS, B, ld))
        k, v, d = (np.ascontiguousarray(a, float) for a in (track.kappa, track.v_ref, track.ds_next))
        ubT = np.ascontiguousarray(track.ub_obstacles if obstacles else track.ub_free)
        lbT = np.ascontiguousarray(track.lb_obstacles if obstacles else track.lb_free)
        wp = np.ascontiguousarray(wp, np.int32)
        x0 = np.ascontiguousarray(x0, float)
        cc = np.ascontiguousarray(cc, float)
        lbp = None if use_table else np.ascontiguousarray(lb, float)
        ubp = None if use_table else np.ascontiguousarray(ub, float)
        rc = self.lib.emu_assemble(C.byref(cfg), C.c_int(k.size), _d(k), _d(v), _d(d), C.c_int(ubT.shape[1]),
                                   _d(ubT), _d(lbT), C.c_int(B), _i(wp), _d(x0), _d(cc), _d(lbp), _d(ubp), _d(qp))
        assert rc == 0
        return qp

    def solve(self, cfg, settings, qp, G=64, want_y=True):
        B = qp.shape[1]
        N = cfg.N
        n, m = 5 * N + 3, 8 * N + 6
        z, u0 = np.zeros((B, n)), np.zeros((B, 2))
        st, it, rs = np.zeros(B, np.int32), np.zeros((B, 2), np.int32), np.zeros((B, 2))
        y = np.zeros((B, m)) if want_y else None
        qp = np.ascontiguousarray(qp)
        if 64 < N + 1 <= 128 and G != 128 and (self.lib.emu_reduced_native(C.byref(cfg), C.byref(settings)) or
                                               self.lib.emu_reduced_native_tt(C.byref(cfg), C.byref(settings))):
            # the launcher's sequence at horizons 64 .. 127: the reduced-native kernel (or its terminal-time twin) with TWO stages
            # per lane in one wavefront, then the workgroup kernel (mode 2) on what it lists (G = 128: the workgroup kernels
            # alone, as mpmpc_set_packing(h, 128))
            yy = y if want_y else np.zeros((B, m))
            nt = C.c_int(0)
            rc = self.lib.emu_solve_rn(C.byref(cfg), C.byref(settings), C.c_int(64), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                                       _d(rs), _d(yy), C.byref(nt))
            assert rc == 0
            ids = np.ascontiguousarray(np.flatnonzero(st == -10), np.int32)          # MPMPC_UNSOLVED
            assert ids.size == nt.value
            if ids.size and self.lib.emu_reduced_native_tail(C.byref(cfg), C.byref(settings)):
                # ... the reduced-native TAIL solver on the same layout first (mpmpc_reduced_tail_pair_kernel<64>)
                ids2, n2 = np.zeros(ids.size, np.int32), C.c_int(0)
                rc = self.lib.emu_solve_rn_tail_pair(C.byref(cfg), C.byref(settings), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                                                     _d(rs), _d(yy), _i(ids), C.c_int(ids.size), _i(ids2), C.byref(n2))
                assert rc == 0
                ids = np.ascontiguousarray(ids2[:n2.value])
            rc = self.wide(128).emuw_solve_tail(C.byref(cfg), C.byref(settings), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                                                _d(rs), _d(yy), _i(ids), C.c_int(ids.size))
            assert rc == 0
            return mpmpc.Solution(z, u0, st, it, rs, y)
        if 128 < N + 1 <= 256 and G != 256 and (self.lib.emu_reduced_native(C.byref(cfg), C.byref(settings)) or
                                                self.lib.emu_reduced_native_tt(C.byref(cfg), C.byref(settings))):
            # horizons 128 .. 255: the reduced-native solver with two stages per lane on a workgroup of 128 lanes, then the
            # general solver on 256 lanes (mode 2) on what it lists (G = 256: the 256-lane workgroup kernels alone, as
            # mpmpc_set_packing(h, 256))
            yy = y if want_y else np.zeros((B, m))
            ids, n = np.zeros(B, np.int32), C.c_int(0)
            tt = bool(self.lib.emu_reduced_native_tt(C.byref(cfg), C.byref(settings)))
            fn = self.wide(128).emuw_solve_rnt_pair if tt else self.wide(128).emuw_solve_rn_pair
            rc = fn(C.byref(cfg), C.byref(settings), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it), _d(rs), _d(yy), _i(ids), C.byref(n))
            assert rc == 0
            ids = np.ascontiguousarray(ids[:n.value])
            if ids.size and not tt and self.lib.emu_reduced_native_tail(C.byref(cfg), C.byref(settings)):
                # ... the tail solver on the same workgroup layout first (mpmpc_reduced_tail_pair_block_kernel)
                ids2, n2 = np.zeros(ids.size, np.int32), C.c_int(0)
                rc = self.wide(128).emuw_solve_rn_tail_pair(C.byref(cfg), C.byref(settings), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                                                            _d(rs), _d(yy), _i(ids), C.c_int(ids.size), _i(ids2), C.byref(n2))
                assert rc == 0
                ids = np.ascontiguousarray(ids2[:n2.value])
            rc = self.wide(256).emuw_solve_tail(C.byref(cfg), C.byref(settings), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                                                _d(rs), _d(yy), _i(ids), C.c_int(ids.size))
            assert rc == 0
            return mpmpc.Solution(z, u0, st, it, rs, y)
        if N + 1 > 64:        # one instance per workgroup of 2 / 4 wavefronts on the device: the wide emulation
            yy = y if want_y else np.zeros((B, m))
            rc = self.wide(mpmpc.stage_ld(N)).emuw_solve(C.byref(cfg), C.byref(settings), _d(qp), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                                                         _d(rs), _d(yy))
            assert rc == 0
            return mpmpc.Solution(z, u0, st, it, rs, y)
        rc = self.lib.emu_solve(C.byref(cfg), C.byref(settings), C.c_int(G), _d(qp), C.c_int(B), _d(z), _d(u0),
                                _i(st), _i(it), _d(rs), _d(y))
        assert rc == 0
        return mpmpc.Solution(z, u0, st, it, rs, y)

    def solve_launch(self, cfg, settings, qp, G=64):
        """what the launcher does: a packed batch (G < 64) runs its early pass packed and its tail one per wave.
        -> (Solution, number of instances handed to the second launch)"""
        B = qp.shape[1]
        N = cfg.N
        n, m = 5 * N + 3, 8 * N + 6
        z, u0, y = np.zeros((B, n)), np.zeros((B, 2)), np.zeros((B, m))
        st, it, rs = np.zeros(B, np.int32), np.zeros((B, 2), np.int32), np.zeros((B, 2))
        nt = C.c_int(0)
        rc = self.lib.emu_solve_launch(C.byref(cfg), C.byref(settings), C.c_int(G), _d(np.ascontiguousarray(qp)), C.c_int(B),
                                       _d(z), _d(u0), _i(st), _i(it), _d(rs), _d(y), C.byref(nt))
        assert rc == 0
        return mpmpc.Solution(z, u0, st, it, rs, y), nt.value

    def solve_rn(self, cfg, settings, qp, G=64, sequential=False):
        """the reduced-native kernel alone (no tail launch); sequential = True: the same kernel with the chain-sequential
        factorisation in place of the cyclic reduction (Solver<..., CR = false>).  -> (Solution, instances left unsolved)"""
        B = qp.shape[1]
        N = cfg.N
        n, m = 5 * N + 3, 8 * N + 6
        z, u0, y = np.zeros((B, n)), np.zeros((B, 2)), np.zeros((B, m))
        st, it, rs = np.zeros(B, np.int32), np.zeros((B, 2), np.int32), np.zeros((B, 2))
        nt = C.c_int(0)
        fn = self.lib.emu_solve_rn_sequential if sequential else self.lib.emu_solve_rn
        rc = fn(C.byref(cfg), C.byref(settings), C.c_int(G), _d(np.ascontiguousarray(qp)), C.c_int(B), _d(z), _d(u0), _i(st), _i(it),
                _d(rs), _d(y), C.byref(nt))
        assert rc == 0
        return mpmpc.Solution(z, u0, st, it, rs, y), nt.value

    def solve_warm(self, cfg, settings, qp, guess, G=64):
        """closed-loop variant: start from the active sets `guess` [B, ld]; -> (Solution, act [B, ld])"""
        B = qp.shape[1]
        N = cfg.N
        n, m = 5 * N + 3, 8 * N + 6
        z, u0, y = np.zeros((B, n)), np.zeros((B, 2)), np.zeros((B, m))
        st, it, rs = np.zeros(B, np.int32), np.zeros((B, 2), np.int32), np.zeros((B, 2))
        act = np.zeros((B, mpmpc.stage_ld(N)), np.int32)
        guess = np.ascontiguousarray(guess, np.int32)
        rc = self.lib.emu_solve_warm(C.byref(cfg), C.byref(settings), C.c_int(G), _d(np.ascontiguousarray(qp)), C.c_int(B),
                                     _i(guess), _d(z), _d(u0), _i(st), _i(it), _d(rs), _d(y), _i(act))
        assert rc == 0
        return mpmpc.Solution(z, u0, st, it, rs, y), act


class EmuBackend:
    """Drop-in for mpmpc.Handle in host-logic tests: same set_path / solve surface, kernels run in
    the CPU lock-step emulation.  Test infrastructure only."""

    class _T:
        pass

    def __init__(self, cfg, settings, emu=None):
        self.cfg, self.settings = cfg, settings
        self.emu = emu or Emul()
        self.t = EmuBackend._T()
        z = np.zeros((2, max(cfg.N, 1)))
        self.t.ub_free = self.t.lb_free = self.t.ub_obstacles = self.t.lb_obstacles = z

    def set_path(self, kappa, v_ref, ds_next):
        self.t.kappa, self.t.v_ref, self.t.ds_next = (np.ascontiguousarray(a, float) for a in (kappa, v_ref, ds_next))

    def solve(self, wp_id, x0, cc_prev, lb=None, ub=None, want_y=False):
        qp = self.emu.assemble(self.cfg, self.t, (np.asarray(wp_id, np.int32), x0, cc_prev, lb, ub))
        return self.emu.solve(self.cfg, self.settings, qp, G=64, want_y=want_y)


class DryHandle(EmuBackend):
    """The RESIDENT surface of mpmpc.Handle (upload / solve_resident / sync / download, set_pipeline / set_outputs / set_packing)
    on the CPU emulation: what `bench.py --dry-run` drives when it rehearses the multi-rank plumbing on a box without GPUs.
    A resident launch is emulated when its results are asked for (download); the launches of a timed loop cost nothing."""

    class _Lib:
        @staticmethod
        def mpmpc_version():
            return b"mpmpc DRY RUN (CPU emulation of the kernels, tests/emul)"

    def __init__(self, cfg, settings, emu=None):
        super().__init__(cfg, settings, emu)
        self.lib = DryHandle._Lib()
        self.pipeline, self.uploaded, self.launches, self._sol = 3, None, 0, None

    def set_packing(self, lanes_per_instance=0):
        pass

    def set_outputs(self, want_y=True):
        pass

    def set_pipeline(self, depth=3):
        self.pipeline = int(depth)

    def upload(self, wp_id, x0, cc_prev, lb=None, ub=None):
        self.uploaded, self._sol = (np.asarray(wp_id, np.int32), x0, cc_prev, lb, ub), None

    def solve_resident(self, B):
        assert self.uploaded is not None and B <= self.uploaded[0].size
        self.launches += 1

    def sync(self):
        pass

    def download(self, B, want_y=False):
        if self._sol is None:
            self._sol = self.solve(*self.uploaded, want_y=want_y)
        return self._sol

    def close(self):
        pass
