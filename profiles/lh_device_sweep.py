"""Ad-hoc: the on-device long-horizon test (tests/test_long_horizon.py::test_long_horizons_on_device) at more horizons - on and
beside the row boundaries of the chains:  python profiles/lh_device_sweep.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "multi-purpose-mpc_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import mpmpc_testlib as T  # noqa: E402
import scenarios  # noqa: E402
import test_long_horizon as TL  # noqa: E402

emu, track = T.Emul(), scenarios.sim_track()
bad = 0
for N in (65, 80, 96, 112, 126, 129, 144, 160, 177, 192, 224, 254):
    for cfgid in (2, 4):
        try:
            TL.test_long_horizons_on_device.__wrapped__(N, cfgid, track, emu) if hasattr(TL.test_long_horizons_on_device, "__wrapped__") else \
                TL.test_long_horizons_on_device(N, cfgid, track, emu)
            print("N %3d config %d: ok" % (N, cfgid))
        except AssertionError as e:
            bad += 1
            print("N %3d config %d: FAILED %s" % (N, cfgid, str(e)[:200]))
print("failures:", bad)
