"""Overlap of kernels in a rocprofv3 --kernel-trace of profiles/overlap_loop.py (profiles/overlap_trace.sh): for the LAST 40 launches
of each solve kernel (the two-handles loop) the share of the tail kernel's run time during which a packed kernel of the other
stream is running too, and the share of wall time with two solve kernels on the chip."""
import csv
import glob
import json
import os
import sys

O, cfg = sys.argv[1], sys.argv[2]
f = max(glob.glob(os.path.join(O, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "mpmpc_reduced_kernel" in r["Kernel_Name"] or "mpmpc_solve_kernel" in r["Kernel_Name"]]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "rn" if "reduced" in r["Kernel_Name"] else "tail", r.get("Stream_Id") or r.get("Queue_Id")) for r in rows))
# the pipelined loop is the last phase of the run: its launches are the last 40 of each kernel
rn = [e for e in ev if e[2] == "rn"][-40:]
t0 = rn[0][0]
win = [e for e in ev if e[0] >= t0]
tails = [e for e in win if e[2] == "tail"]
rns = [e for e in win if e[2] == "rn"]


def overlap(a, bs):
    tot = 0
    for b in bs:
        lo, hi = max(a[0], b[0]), min(a[1], b[1])
        if hi > lo and b[3] != a[3]:
            tot += hi - lo
    return tot


tail_time = sum(e[1] - e[0] for e in tails)
tail_ov = sum(min(overlap(e, rns), e[1] - e[0]) for e in tails)
span = max(e[1] for e in win) - t0
busy = sum(e[1] - e[0] for e in win)
out = ["config %s, profiles/overlap_loop.py under rocprofv3 --kernel-trace (40 steps on two handles in turn, nothing else)" % cfg,
       open(os.path.join(O, "line.txt")).read().strip().splitlines()[-1],
       "launches in the window: %d packed kernels, %d tail kernels, streams / queues seen: %s" % (len(rns), len(tails), sorted({e[3] for e in win})),
       "tail kernels: %.1f us each on average, %.0f %% of their run time beside a packed kernel of the other handle" % (tail_time / max(len(tails), 1) / 1e3, 100.0 * tail_ov / max(tail_time, 1)),
       "window %.1f us, sum of kernel durations %.1f us: %.2f solve kernels on the chip on average" % (span / 1e3, busy / 1e3, busy / span)]
print("\n".join(out))
open(os.path.join(O, "overlap_cfg%s.txt" % cfg), "w").write("\n".join(out) + "\n")
