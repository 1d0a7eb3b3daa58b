"""Flop census of K2 from the counting build of the lock-step emulation (tests/_build/libmpmpc_emul_count.so,
-DMPMPC_COUNT_OPS; FMA = 2, add / mul / div / sqrt / rcp seed = 1; compares, selects, lane moves not counted).

Two counts per solve, both from the SAME executed code (mpmpc_core.hpp marks the contexts, lane_emu.hpp counts):

  algorithmic   the structure-exploiting count SURVEY 8(d) asks for: lane-parallel instructions count for the lanes
                that hold a stage (N + 1; in the split layout of the interior point N + 1 state lanes + N input lanes
                with 2 of their 3 entries), and every serial sweep of the twisted factorisation / substitution counts
                ONE step per stage (each stage's step does useful work once per sweep, although all lanes execute
                every step);
  executed      round 1's figure: every wave instruction times the N + 1 stage-holding lanes, serial steps included
                as often as they are executed.

Fitted per solve at the default settings as  flops = c0 + c1 * ipm_iters  (the one ADMM iteration, the Ruiz passes,
the active-set rounds and the certificate are in c0 / amortised in c1) and printed with the per-stage cost of the two
linear-algebra pieces (factor, KKT solve).  bench.py carries the fitted constants.  Runs in the authoring container:
    python profiles/census.py [config] [instances] [reduce (1 | 0)]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("multi-purpose-mpc_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import mpmpc                    # noqa: E402
import mpmpc_testlib as tl      # noqa: E402
import scenarios                # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
emu = tl.Emul()
cnt = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmpmpc_emul_count.so"))
emu.lib = cnt
tr = scenarios.sim_track()
sc = scenarios.make(config, tr, B)
N = sc.N
cfg = tl.stock_config(N, weights=sc.weights)
st = mpmpc.default_settings(reduce=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
out = (C.c_double * 4)()
rows = []
for i in range(B):
    inp = (sc.wp_id[i:i + 1], sc.x0[i:i + 1], sc.cc_prev[i:i + 1], sc.lb[i:i + 1], sc.ub[i:i + 1])
    qp = emu.assemble(cfg, tr, inp, obstacles=sc.obstacles)
    cnt.emu_op_flops(out, 1)
    sol, _ = emu.solve_launch(cfg, st, qp, G=64)          # the launcher's own sequence: reduced-native kernel, then its tail
    cnt.emu_op_flops(out, 1)
    par, split, ser_exec, ser_one = out[0], out[1], out[2], out[3]
    algorithmic = (N + 1) * par + ((N + 1) + 2.0 * N / 3.0) * split + (N + 1) * ser_one
    executed = (N + 1) * (par + split + ser_exec)
    rows.append((sol.status[0], sol.iters[0, 0], sol.iters[0, 1], algorithmic, executed))
r = np.array(rows, float)
for name, col in (("algorithmic", 3), ("executed", 4)):
    for stt in sorted(set(r[:, 0])):
        m = r[:, 0] == stt
        if m.sum() < 4:
            continue
        A = np.stack([np.ones(m.sum()), r[m, 2]], axis=1)
        coef, *_ = np.linalg.lstsq(A, r[m, col], rcond=None)
        err = (A @ coef - r[m, col]) / r[m, col]
        print("config %d (N = %d), status %2d, %-11s flops per solve = %9.0f + %8.0f * ipm_iters   (rms error %.1f %%, mean %.3f MFLOP, "
              "%d solves, ipm iterations %.1f)" % (config, N, stt, name, coef[0], coef[1], 100 * np.sqrt(np.mean(err ** 2)),
                                                  r[m, col].mean() / 1e6, m.sum(), r[m, 2].mean()))
if N + 1 <= 32:
    cnt.emu_census_pieces(N, out)
    print("per stage: factor %.0f flops (%.0f lane-parallel + %.1f in its serial step), KKT solve %.0f (%.0f + %.1f in the "
          "two sweep steps)" % (out[0] + out[1], out[0], out[1], out[2] + out[3], out[2], out[3]))
np.save(os.path.join("/tmp", "census_%d.npy" % config), r)
