"""`osqp`-shaped stand-in used ONLY while generating golden fixtures.

The reference imports a module called `osqp` (src/MPC.py:2, src/reference_path.py:7)
that does not exist in this container.  This stand-in records the exact
(P, q, A, l, u) handed to `setup()` and answers `solve()` with the oracle's
KKT-certified optimum (oracle/osqp_np.py), or with a None-filled vector when the
problem is primal infeasible, which is what stock OSQP hands back and what makes
the reference's bare `except:` (src/MPC.py:208) fire.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "oracle"))
import osqp_np  # noqa: E402

CAPTURES = []          # every setup() call appends a dict here
SOLVE = True           # set False to skip the (slow, dense) solve and return None-filled x
SETTINGS = osqp_np.Settings(polish=2)     # stock OSQP defaults + certified polish


class _Res:
    pass


class OSQP:
    def __init__(self):
        self.cap = None

    def setup(self, P=None, q=None, A=None, l=None, u=None, **kw):
        self.cap = dict(P=P.tocsc().copy(), q=np.array(q, float), A=A.tocsc().copy(),
                        l=np.array(l, float), u=np.array(u, float), kw=dict(kw))
        CAPTURES.append(self.cap)

    def solve(self):
        r = _Res()
        n = self.cap["q"].size
        if not SOLVE:
            r.x = np.array([None] * n)
            r.status = "skipped"
            return r
        c = self.cap
        res = osqp_np.solve(c["P"].toarray(), c["q"], c["A"].toarray(), c["l"], c["u"], SETTINGS)
        c["res"] = res
        r.status_val = res.status
        if res.status in (osqp_np.SOLVED, osqp_np.SOLVED_INACCURATE, osqp_np.MAX_ITER_REACHED):
            r.x = res.x
            r.y = res.y
        else:
            r.x = np.array([None] * n)
            r.y = np.array([None] * c["l"].size)
        return r
