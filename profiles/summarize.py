"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<round>/) into the committed
summaries under profiles/<round>/:  python profiles/summarize.py r1"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(root, "gpurun_out", rnd), os.path.join(root, "profiles", rnd)
os.makedirs(P, exist_ok=True)
for name in ("bench_cfg2.json", "bench_cfg3.json", "bench_cfg4.json", "bench_cfg5.json", "bench_cfg4_strict.json", "bench_cfg5_strict.json",
             "bench_cfg2_b65536.json", "host.txt"):
    if os.path.exists(os.path.join(O, name)):
        shutil.copy(os.path.join(O, name), os.path.join(P, name))
if os.path.exists(os.path.join(O, "trace_b65536.json")) and not os.path.exists(os.path.join(O, "bench_cfg2_b65536.json")):
    shutil.copy(os.path.join(O, "trace_b65536.json"), os.path.join(P, "bench_cfg2_b65536.json"))
for src, dst in (("trace", "bench_cfg2_kernel_stats.csv"), ("trace_b65536", "bench_cfg2_b65536_kernel_stats.csv"),
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
             "pmc_sq1_b65536", "pmc_sq2_b65536"):
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
for src, dst in (("trace_cfg4", "bench_cfg4_profiled.json"), ("trace_cfg5", "bench_cfg5_profiled.json")):
    if os.path.exists(os.path.join(O, src + ".json")):
        shutil.copy(os.path.join(O, src + ".json"), os.path.join(P, dst))
json.dump(out, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
