"""Per-step cost of the single-car drop-in MPC.get_control() with the corridor on the host (as the reference)
and on the device, profiled:  python profiles/latency_get_control.py  (on the GPU box)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("multi-purpose-mpc_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import test_host_mpc as H   # noqa: E402

for mode in ("host", "device"):
    m, rp, car = H.build_world()
    mpc = H.make_mpc(car, 30, corridor=mode)
    for _ in range(5):
        car.drive(mpc.get_control())
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    for _ in range(100):
        car.drive(mpc.get_control())
    pr.disable()
    print("corridor=%s: %.3f ms per get_control() + drive()" % (mode, (time.perf_counter() - t) * 10))
    if "-v" in sys.argv:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
