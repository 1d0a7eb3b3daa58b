"""Step A of VERDICT r5 item 1: instruction census of the reduced-native solver in the layout with TWO stages per lane
(lane_pair.hpp: 16 lanes per instance at N = 30, four instances per wavefront) against the shipped one-stage layout
(<32,16>: two instances per wavefront), from the counting build of the lock-step emulation (tests/_build/
libmpmpc_emul_count.so: wave-level operations by class - FMA, add / mul, reciprocal, reciprocal square root, compare / select /
max, lane shift, wave reduction).  The classes are weighted to VALU instructions with the instruction costs of the device
backend (lane_gpu.hpp) and the total is calibrated on the PMC count of the shipped kernel (13 581 SQ_INSTS_VALU per packed
wave on config 2, profiles/r5/pmc_wait_r5w.json); go / no-go: >= 25 % fewer VALU instructions PER INSTANCE.

    python profiles/census_s2.py [config] [B]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("multi-purpose-mpc_amd", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
import mpmpc                    # noqa: E402
import mpmpc_testlib as tl      # noqa: E402
import scenarios                # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
emu = tl.Emul()
emu.lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmpmpc_emul_count.so"))
tr = scenarios.sim_track()
sc = scenarios.make(config, tr, B)
cfg = tl.stock_config(sc.N, weights=sc.weights)
st = mpmpc.default_settings()
qp = emu.assemble(cfg, tr, (sc.wp_id, sc.x0, sc.cc_prev, sc.lb, sc.ub), obstacles=sc.obstacles)
names = ("fma", "addmul", "rcp", "rsqrt", "cmpsel", "shift", "reduce")
# VALU instructions per wave-level operation of a class on the device (lane_gpu.hpp): rcp_ = seed + 3 FMA, rsqrt_ = seed + mul + 3
# FMA + mul, a select = two v_cndmask (a max / min / compare one instruction: 1.6 on the mix), a lane shift = one DPP move per dword,
# a wave reduction = log2(lanes of the butterfly) steps of two moves + one operation
W = {32: dict(fma=1, addmul=1, rcp=4, rsqrt=6, cmpsel=1.6, shift=2, reduce=15), 16: dict(fma=1, addmul=1, rcp=4, rsqrt=6, cmpsel=1.6, shift=2, reduce=12)}
out = (C.c_longlong * 7)()
res = {}
for G, per_wave in ((32, 2), (16, 4)):
    emu.lib.emu_op_count(out, 1)
    sol, nt = emu.solve_rn(cfg, st, qp, G=G)
    emu.lib.emu_op_count(out, 1)
    waves = (B + per_wave - 1) // per_wave
    c = {n: out[i] / waves for i, n in enumerate(names)}
    res[G] = dict(per_wave=c, waves=waves, weighted=sum(W[G][n] * c[n] for n in names), ipm=sol.iters[:, 1].mean(), tail=nt)
cal = 13581.0 / res[32]["weighted"] if config == 2 else 1.0
print("config %d, B = %d, N = %d; wave-level operations per wave by class" % (config, B, sc.N))
for G, per_wave in ((32, 2), (16, 4)):
    r = res[G]
    print("%s: %d waves, %s" % ("one stage per lane <32,16>, 2 instances / wave" if G == 32 else "two stages per lane, 16 lanes, 4 instances / wave", r["waves"],
                                ", ".join("%s %.0f" % (n, r["per_wave"][n]) for n in names)))
    print("    weighted VALU estimate per wave %.0f, per instance %.0f (calibration x %.3f: %.0f per wave, %.0f per instance); ipm iterations %.3f, tail %d"
          % (r["weighted"], r["weighted"] / per_wave, cal, cal * r["weighted"], cal * r["weighted"] / per_wave, r["ipm"], r["tail"]))
a, b = res[32]["weighted"] / 2, res[16]["weighted"] / 4
print("VALU per instance: two stages per lane / one stage per lane = %.3f  (%.1f %% fewer)" % (b / a, 100 * (1 - b / a)))
