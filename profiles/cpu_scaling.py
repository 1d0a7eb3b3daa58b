"""Thread scaling of the CPU baseline (oracle/osqp_port.c) on this host: python profiles/cpu_scaling.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("multi-purpose-mpc_amd", "oracle", "tests", "."):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle_c     # noqa: E402
import scenarios    # noqa: E402

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
tr = scenarios.sim_track()
sc = scenarios.make(2, tr, 1024)
limits = dict(umin=scenarios.UMIN, umax=scenarios.UMAX, xmin=scenarios.XMIN, xmax=scenarios.XMAX, ay_max=scenarios.AY_MAX,
              wheelbase=scenarios.CAR_LENGTH)
nt = 1
while nt <= os.cpu_count():
    base, _ = oracle_c.timed_baseline(tr, sc, scenarios.WEIGHTS[sc.weights], limits, seconds=2.0, nthreads=nt)
    print("threads %3d: %8.0f solves/s  (%.0f per thread)" % (nt, base["value"], base["value"] / nt))
    nt *= 2
