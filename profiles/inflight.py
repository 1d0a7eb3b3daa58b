"""How many solve kernels are on the chip, and where the streams wait: from a rocprofv3 --kernel-trace of bench.py.
    python profiles/inflight.py <dir with *kernel_trace.csv> [skip_launches]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 400
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "mpmpc_reduced" in r["Kernel_Name"] or "mpmpc_solve_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:skip + 2000]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("%d kernels over %.3f ms: %.2f in flight on average, %.2f us per kernel start, mean duration %.1f us" %
      (len(rows), (t1 - t0) * 1e-6, busy / (t1 - t0), (t1 - t0) * 1e-3 / len(rows), busy * 1e-3 / len(rows)))
per = collections.defaultdict(list)
for r in rows:
    per[(r.get("Queue_Id"), r.get("Stream_Id"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in sorted(per.items()):
    gaps = [(b[0] - a[1]) * 1e-3 for a, b in zip(v, v[1:])]
    print("queue / stream %s: %d kernels, gap between consecutive kernels mean %.1f us (min %.1f, max %.1f)" %
          (k, len(v), sum(gaps) / max(1, len(gaps)), min(gaps or [0]), max(gaps or [0])))
