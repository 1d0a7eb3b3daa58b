"""Which branch of src/MPC.py:185-216 does get_control take - a fresh plan or the fallback - with the DEFAULT settings of the
device, against the restated stock OSQP (the C oracle at OSQP's defaults: the arithmetic of the reference's own solver call,
src/MPC.py:159,183)?  Lists every instance of configs 4 / 5 on which the two disagree: phase 1's least violation (resid[0] of
the device), OSQP's status / iteration count / primal residual.  VERDICT r4 item 3.

    python profiles/branch_agreement.py [config ...] > profiles/r5/branch_agreement.txt        (on a GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "multi-purpose-mpc_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

import mpmpc  # noqa: E402
import mpmpc_testlib as T  # noqa: E402
import scenarios  # noqa: E402


compare = T.branch_compare          # (tests/mpmpc_testlib.py: the GPU test pins what this script prints)


if __name__ == "__main__":
    print("# %s" % mpmpc.load_library().mpmpc_version().decode())
    for c in [int(a) for a in sys.argv[1:]] or [4, 5]:
        for B in ((8192,) if c == 4 else (8192, 65536)):
            r = compare(c, B)
            print("config %d, B = %d: device %s, restated stock OSQP %s" % (c, B, r["device"], r["stock"]))
            print("  same branch on %d of %d instances (%.5f); OSQP's primal tolerance at the steering limit: %.4e" % (B - len(r["rows"]), B, r["agreement"], r["threshold"]))
            print("  instance  device  least violation   stock  iterations  pri_res      why")
            for i, ds, dv, ss, si, sp in r["rows"]:
                why = "OSQP abandoned at max_iter: neither of its tests passed, it returns its iterate" if si >= 4000 else \
                      ("within %.2f %% of the tolerance: OSQP stopped an iterate short of its limit" % (100 * abs(dv / r["threshold"] - 1)) if abs(dv / r["threshold"] - 1) < 0.05 else "?")
                print("  %8d  %6d  %.4e        %5d  %10d  %.4e   %s" % (i, ds, dv, ss, si, sp, why))
