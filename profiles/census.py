"""Instruction census of K2 from the counting build of the lock-step emulation
(tests/_build/libmpmpc_emul_count.so, -DMPMPC_COUNT_OPS) and a least-squares fit of
    FP64 flops per lane = c0 + c1 * admm_iters + c2 * ipm_iters        (FMA = 2, add/mul/div/sqrt = 1)
that bench.py's `roofline_fp64` multiplies by the N + 1 lanes holding a stage.  Runs in the authoring
container (no GPU):  python profiles/census.py [configs, e.g. 2,4] [instances per config]"""
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

configs = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 48
emu = tl.Emul()
cnt = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmpmpc_emul_count.so"))
emu.lib = cnt
tr = scenarios.sim_track()
rows, rhs, mix, dflt = [], [], [], []
out = (C.c_longlong * 7)()
N = None
for config in configs:
    sc = scenarios.make(config, tr, B)
    assert N in (None, sc.N), "pool configurations of one horizon only"
    N = sc.N
    cfg = tl.stock_config(sc.N, weights=sc.weights)
    # pure ADMM runs anchor the setup and per-iteration terms, the polished ones the interior-point term
    for kw in (dict(polish=0), dict(polish=0, eps_abs=1e-4, eps_rel=1e-4), dict(), dict(early_polish=0)):
        st = mpmpc.default_settings(**kw)
        for i in range(B):
            inp = (sc.wp_id[i:i + 1], sc.x0[i:i + 1], sc.cc_prev[i:i + 1], sc.lb[i:i + 1], sc.ub[i:i + 1])
            qp = emu.assemble(cfg, tr, inp, obstacles=sc.obstacles)
            cnt.emu_op_count(out, 1)
            sol = emu.solve(cfg, st, qp, G=64)
            cnt.emu_op_count(out, 1)
            if sol.status[0] != 1:
                continue
            rows.append([1.0, float(sol.iters[0, 0]), float(sol.iters[0, 1])])
            rhs.append(2.0 * out[0] + out[1] + out[2] + out[3])
            mix.append(list(out))
            dflt.append(not kw)
A, b, mix, dflt = np.array(rows), np.array(rhs), np.array(mix, float), np.array(dflt)
from scipy.optimize import nnls                                                # relative least squares, coefficients >= 0
coef, _ = nnls(A / b[:, None], np.ones_like(b))
res = (A @ coef - b) / b
print("configs %s (N=%d): flops per lane = %.4g + %.4g * admm_iters + %.4g * ipm_iters   (rms error %.1f %%, at the "
      "defaults %.1f %% with bias %+.1f %%; %d solves)" % (configs, N, coef[0], coef[1], coef[2], 100 * np.sqrt(np.mean(res ** 2)),
      100 * np.sqrt(np.mean(res[dflt] ** 2)), 100 * np.mean(res[dflt]), len(b)))
print("wave instructions per solve at the default settings (mean): fma %.0f  add/mul %.0f  div %.0f  sqrt %.0f  "
      "cmp/sel %.0f  lane shifts %.0f  reductions %.0f" % tuple(mix[dflt].mean(axis=0)))
