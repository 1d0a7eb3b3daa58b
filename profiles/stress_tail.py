"""Randomised sweep of the reduced-native TAIL solver against the general kernel's tail, in the CPU emulation (no GPU):
    python profiles/stress_tail.py [seed]
60 trials of random horizons (3 .. 31), configurations (2 / 4 / 5), batch sizes, packings, both verdict semantics, every third
trial with squeezed corridors: statuses must be identical, points / multipliers within 1e-9 (they are within 4.4e-16).
TAIL_MODE=2 (default here): the tail solver's one-instance-per-wave form; TAIL_MODE=1: its default form, two instances per wave
(points of infeasible / marginal instances then agree to ~1e-7 only: pass a looser bound by eye)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), ROOT, os.path.join(ROOT, "multi-purpose-mpc_amd")]
import numpy as np
import mpmpc, mpmpc_testlib as T, scenarios
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tr = scenarios.sim_track(); em = T.Emul()
tot = tails = left2 = 0; worst = 0.0; bad = 0
for trial in range(60):
    N = int(rng.choice([3, 4, 7, 10, 15, 16, 17, 24, 30, 31]))
    cfg_id = int(rng.choice([2, 4, 4, 5]))
    B = int(rng.integers(8, 400))
    sc = scenarios.make(cfg_id, tr, B=B, N=N)
    perm = rng.permutation(B)
    wp, x0, cc, lb, ub = sc.wp_id[perm], sc.x0[perm], sc.cc_prev[perm], sc.lb[perm], sc.ub[perm]
    x0 = x0 + rng.normal(0, 0.02, x0.shape) * np.array([1.0, 1.0, 0.0])
    # squeeze some corridors so that marginal / infeasible cases appear at every horizon
    if trial % 3 == 0:
        mid = 0.5 * (lb + ub); half = 0.5 * (ub - lb) * rng.uniform(0.02, 1.0, size=(B, 1))
        lb, ub = mid - half, mid + half
    cfg = T.stock_config(N, sc.weights, max_batch=B)
    st = mpmpc.default_settings(phase1_accept=trial % 2)
    qp = em.assemble(cfg, tr, (wp, x0, cc, lb, ub))
    G = int(rng.choice([g for g in (64, 32, 16) if N + 1 <= g]))
    em.lib.emu_set_lean_tail(0); a, nt = em.solve_launch(cfg, st, qp, G=G)
    em.lib.emu_set_lean_tail(int(os.environ.get("TAIL_MODE", "2"))); b, nt2 = em.solve_launch(cfg, st, qp, G=G)
    l2 = em.lib.emu_last_tail2()
    same = np.array_equal(a.status, b.status)
    dz = float(np.abs(a.z - b.z).max()); dy = float((np.abs(a.y - b.y) / np.maximum(1.0, np.abs(a.y).max(axis=1, keepdims=True))).max())
    dr = float(np.abs(a.resid - b.resid).max())
    worst = max(worst, dz, dy)
    tot += B; tails += nt; left2 += l2
    flag = "" if same and dz < 1e-9 and dy < 1e-9 else "  <-- MISMATCH"
    bad += bool(flag)
    print("trial %2d cfg %d N %2d B %3d G %2d accept %d: tail %3d left %2d status %s dz %.1e dy %.1e dresid %.1e st %s%s" % (
        trial, cfg_id, N, B, G, trial % 2, nt, l2, same, dz, dy, dr, dict(zip(*np.unique(b.status, return_counts=True))), flag), flush=True)
print("instances %d, tail %d, left to the general kernel %d, worst %.2e, mismatching trials %d" % (tot, tails, left2, worst, bad))
