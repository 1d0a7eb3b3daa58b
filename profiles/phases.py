"""Where K2's time goes: per-wave phase clocks of a profiling build (-DMPMPC_PHASE_CLOCK, see
mpmpc_hip.hip / mpmpc_core.hpp MPMPC_TICK_*).  Usage on the GPU box:

    python profiles/phases.py [config] [batch]

with profiles/_ab/P.so built by
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMPMPC_PHASE_CLOCK \
          -Iinclude -o profiles/_ab/P.so multi-purpose-mpc_amd/csrc/mpmpc_hip.hip
Prints mean / max over the waves of each phase in microseconds (wall_clock64, 10 ns ticks)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-purpose-mpc_amd"))
sys.path.insert(0, ROOT)
import mpmpc      # noqa: E402
import scenarios  # noqa: E402

NAMES = {0: "load", 1: "ruiz scaling", 2: "admm", 3: "polish (all)", 4: "  interior point", 5: "  active set",
         6: "  certificate", 9: "phase 1", 7: "store", 8: "kernel body", 10: "    ipm residuals", 11: "    ipm rcp + factor",
         12: "    ipm kkt solves", 13: "    as factor", 14: "    as kkt solves"}
COUNTS = {16: "ipm iterations", 17: "as rounds", 18: "as solves"}


def main():
    cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    lib = mpmpc.load_library(os.path.join(ROOT, "profiles", "_ab", "P.so"))
    lib.mpmpc_debug_phase.argtypes = [C.c_void_p, C.c_int]
    mpmpc._lib = lib            # the handles below run in the profiling build
    tr = scenarios.sim_track()
    n_over = int(os.environ.get("MPMPC_PHASES_N", "0"))       # another horizon (above 50 the corridor tables come from the emulation)
    if n_over:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mpmpc_testlib as T
        tw = T.wide_track(tr, T.Emul(), max(n_over, 50))
        sc = scenarios.make(cfg_id, tw, B=int(sys.argv[2]) if len(sys.argv) > 2 else None, N=n_over)
    else:
        sc = scenarios.make(cfg_id, tr, B=int(sys.argv[2]) if len(sys.argv) > 2 else None)
    Q, R, QN = scenarios.WEIGHTS[scenarios.CONFIGS[cfg_id]["weights"]]
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX,
                            scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=sc.B)
    # MPMPC_PHASES_SET="key=value,key=value": solver settings away from the defaults (tolerance sweeps)
    over = {}
    for kv in filter(None, os.environ.get("MPMPC_PHASES_SET", "").split(",")):
        k, v = kv.split("=")
        over[k] = float(v) if any(c in v for c in ".eE") else int(v)
    h = mpmpc.Handle(cfg, mpmpc.default_settings(**over))
    h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
    # one launch in flight and the packing of such a handle, unless MPMPC_PHASES_PIPELINE says otherwise: the per-wave
    # accumulators of the profiling build are indexed by block, so launches must not overlap; PIPELINE > 1 selects the
    # throughput packing (two instances per wave from 128 instances on) and the launches are separated by syncs below
    depth = int(os.environ.get("MPMPC_PHASES_PIPELINE", "1"))
    h.set_pipeline(depth)
    h.upload(sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub)
    for _ in range(3):
        h.solve_resident(sc.B)
    h.sync()
    buf = np.zeros((4096, 32), np.int64)
    lib.mpmpc_debug_phase(None, 1)
    reps = 5
    for _ in range(reps):
        h.solve_resident(sc.B)
        h.sync()
    lib.mpmpc_debug_phase(buf.ctypes.data, 0)
    waves = int((buf[:, 8] != 0).sum()) or int((buf[:, 3] != 0).sum())       # (the workgroup kernel has no body clock)
    t = buf[:waves].astype(np.float64) / reps
    print("config %d  B=%d N=%d  waves=%d" % (cfg_id, sc.B, sc.N, waves))
    for i, name in NAMES.items():
        print("%-24s mean %8.2f us   max %8.2f us" % (name, t[:, i].mean() * 0.01, t[:, i].max() * 0.01))
    for i, name in COUNTS.items():
        print("%-24s mean %8.2f      max %8.0f" % (name, t[:, i].mean(), t[:, i].max()))
    # what the slowest waves spend their time on (the launch ends with them)
    order = np.argsort(-t[:, 8])[:8]
    print("slowest waves: body us | ipm its, us | as rounds, solves, us")
    for w in order:
        print("  wave %5d: %7.2f | %2.0f %7.2f | %1.0f %2.0f %6.2f" % (w, t[w, 8] * 0.01, t[w, 16], t[w, 4] * 0.01, t[w, 17], t[w, 18], t[w, 5] * 0.01))
    q = np.percentile(t[:, 8], [50, 90, 99, 100]) * 0.01
    print("body percentiles 50/90/99/100: %.1f %.1f %.1f %.1f us" % tuple(q))


if __name__ == "__main__":
    main()
