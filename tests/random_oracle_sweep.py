"""Randomised sweep: the lock-step emulation of the kernels (all lane packings, packed pass + tail launch) against the
C port of the oracle on random (configuration, horizon 3..50, batch) draws - every status, every control to 1e-6, no
ADMM fallback, no alternative optimum.  Not collected by pytest (minutes per hundred trials):
    python tests/random_oracle_sweep.py SEED TRIALS"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, d) for d in ("tests", "oracle", "multi-purpose-mpc_amd")] + [ROOT]
import numpy as np
import mpmpc, mpmpc_testlib as T, scenarios, oracle_c
tr = scenarios.sim_track()
emu = T.Emul()
rng = np.random.default_rng(int(sys.argv[1]))
bad = 0; tot = 0
for trial in range(int(sys.argv[2])):
    cid = int(rng.choice([2, 3, 4])); N = int(rng.integers(3, 51)); B = int(rng.integers(16, 96)); G = 64 if N + 1 > 32 else int(rng.choice([64, 32] if N + 1 > 16 else [64, 32, 16]))
    sc = scenarios.make(cid, tr, B=B, N=N)
    cfg = T.stock_config(N, sc.weights, max_batch=B)
    qp = emu.assemble(cfg, tr, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub))
    sol, n_tail = emu.solve_launch(cfg, mpmpc.default_settings(phase1_accept=0), qp, G=G)      # (the oracle reports every proven infeasibility)
    cfgc = oracle_c.mpc_cfg(N, scenarios.WEIGHTS[sc.weights], scenarios.UMIN, scenarios.UMAX, scenarios.XMIN, scenarios.XMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH)
    out = oracle_c.mpc_batch(cfgc, oracle_c.settings(), tr.kappa, tr.v_ref, tr.ds_next, sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub, 8, want_y=True)
    tot += B
    same = np.array_equal(sol.status, out['status'])
    worst, alt = T.controls_vs_reference(qp, N, sol, out, 1e-6)
    admm = (sol.iters[:, 0] > 1).sum(), (out['iters'][:, 0] > 1).sum()
    if not same or worst > 1e-6 or len(alt) or admm[0] or admm[1]:
        bad += 1
        d = np.flatnonzero(sol.status != out['status'])
        print('trial', trial, 'cfg', cid, 'N', N, 'B', B, 'G', G, 'status same', same, d[:5], sol.status[d][:5], out['status'][d][:5], 'worst du %.1e' % worst, 'alt', alt, 'admm>1 (emu, C)', admm, flush=True)
print('trials', int(sys.argv[2]), 'instances', tot, 'flagged', bad)
