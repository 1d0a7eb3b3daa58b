"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<round>/) into the committed
summaries under profiles/<round>/:  python profiles/summarize.py r1"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(root, "gpurun_out", rnd), os.path.join(root, "profiles", rnd)
os.makedirs(P, exist_ok=True)
for name in ("bench_cfg2.json", "bench_cfg3.json", "bench_cfg4.json", "bench_cfg5.json", "bench_cfg4_strict.json", "bench_cfg5_strict.json",
             "bench_cfg2_b65536.json", "host.txt"):
    if os.path.exists(os.path.join(O, name)):
        shutil.copy(os.path.join(O, name), os.path.join(P, name))
if os.path.exists(os.path.join(O, "trace_b65536.json")) and not os.path.exists(os.path.join(O, "bench_cfg2_b65536.json")):
    shutil.copy(os.path.join(O, "trace_b65536.json"), os.path.join(P, "bench_cfg2_b65536.json"))
for src, dst in (("trace", "bench_cfg2_kernel_stats.csv"), ("trace_driver", "bench_cfg2_driver_command_kernel_stats.csv"),
                 ("trace_b65536", "bench_cfg2_b65536_kernel_stats.csv"),
                 ("trace_cfg3", "bench_cfg3_kernel_stats.csv"), ("trace_cfg4", "bench_cfg4_kernel_stats.csv"), ("trace_cfg5", "bench_cfg5_kernel_stats.csv")):
    fs = glob.glob(os.path.join(O, src, "*", "*kernel_stats.csv"))
    if fs:                                   # gpurun merges runs into the same directory: newest wins
        shutil.copy(max(fs, key=os.path.getmtime), os.path.join(P, dst))
out = {}
# the library the counters were read on: bench.py reports `traffic` only for a library built from the same sources
try:
    ver = json.load(open(os.path.join(O, "trace.json")))["library"]
    out["library"] = ver
    out["library_source_hash"] = ver.split("src ")[1].rstrip(")")
except Exception as e:
    print("no library version in the profiled bench line:", e)
for name in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_fetch_b65536", "pmc_write_b65536", "pmc_sq1_cfg4", "pmc_sq2_cfg4",
             "pmc_sq1_b65536", "pmc_sq2_b65536", "pmc_fetch_cfg3", "pmc_write_cfg3", "pmc_fetch_cfg4", "pmc_write_cfg4", "pmc_fetch_cfg5",
             "pmc_write_cfg5"):
    fs = glob.glob(os.path.join(O, name, "*", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "copy" not in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        out.setdefault(name, {}).setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
for src, dst in (("trace_cfg4", "bench_cfg4_profiled.json"), ("trace_cfg5", "bench_cfg5_profiled.json"),
                 ("trace_driver", "bench_cfg2_driver_command_profiled.json")):
    if os.path.exists(os.path.join(O, src + ".json")):
        shutil.copy(os.path.join(O, src + ".json"), os.path.join(P, dst))
# ---- the TIMED launches of every traced bench run: rows of the kernel trace in start order, without the
# prewarm + warm-up launches at the head and the profile / isolated launches bench.py issues after its timed loop; per kernel the
# average duration and, over the window, the average number of solve kernels on the chip (launches are double-buffered)
timed = {}
for src, wl in (("trace", "bench_cfg2"), ("trace_driver", "bench_cfg2_driver_command"), ("trace_b65536", "bench_cfg2_b65536"), ("trace_cfg3", "bench_cfg3"), ("trace_cfg4", "bench_cfg4"),
                ("trace_cfg5", "bench_cfg5")):
    fs = glob.glob(os.path.join(O, src, "*", "*kernel_trace.csv"))
    try:
        line = json.load(open(os.path.join(O, src + ".json")))
    except Exception:
        continue
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(max(fs, key=os.path.getmtime)))
            if "mpmpc_reduced" in r["Kernel_Name"] or "mpmpc_solve_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one per step: the step's first kernel (not the tail kernels, which carry "reduced" in their name as well)
    first = [r for r in rows if "mpmpc_reduced_kernel" in r["Kernel_Name"] or "mpmpc_reduced_t_kernel" in r["Kernel_Name"]] or rows
    skip, n = int(line.get("prewarm", 300)) + int(line["warmup"]), int(line["steps"]) * int(line.get("repeats", 1))
    win = first[skip:skip + n]
    if len(win) < n:
        print("trace of %s too short: %d launches after %d skipped, %d expected" % (wl, len(win), skip, n))
        continue
    t0, t1 = int(win[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in win)
    inside = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
    per = collections.defaultdict(list)
    for r in inside:
        per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in inside)
    timed[wl] = {"library": line.get("library"), "steps": int(line["steps"]), "repeats": int(line.get("repeats", 1)), "launches": n,
                 "kernels": {k: {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)} for k, v in per.items()},
                 "streams": sorted({r["Stream_Id"] for r in inside}), "solve_kernels_in_flight_avg": busy / float(t1 - t0),
                 "ms_per_step_in_trace": (t1 - t0) * 1e-6 / n, "ms_per_step_reported": line["ms_per_step"]}
json.dump(timed, open(os.path.join(P, "kernel_timed.json"), "w"), indent=1)
print("timed launches:", json.dumps(timed, indent=1)[:1500])
json.dump(out, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
