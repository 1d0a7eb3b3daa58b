"""Condense the raw rocprofv3 --pmc output of profiles/collect_wait.sh (gpurun_out/<tag>/) into
profiles/<round>/pmc_wait_<tag>.json: per workload and kernel the mean of every counter over the launches, the
kernel-trace averages, and the derived shares (quad-cycle counters as fractions of SQ_WAVE_CYCLES).

    python profiles/summarize_wait.py r3a [r3]
"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r3a"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", rnd)
os.makedirs(P, exist_ok=True)


def short(k):
    return k.split("(")[0].replace("void ", "")


out = {}
for js in sorted(glob.glob(os.path.join(O, "*_trace.json"))):
    try:
        out["library"] = json.load(open(js))["library"]
        break
    except Exception:
        pass
for d in sorted(glob.glob(os.path.join(O, "*"))):
    if not os.path.isdir(d):
        continue
    wl, grp = os.path.basename(d).split("_", 1)
    fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if fs:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            k = short(r["Kernel_Name"])
            if "mpmpc" in k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            out.setdefault(wl, {}).setdefault(k, {})[c] = sum(v) / len(v)
            out[wl][k].setdefault("_launches", {})[grp] = len(v)
    fs = glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))
    if fs:
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
            k = short(r["Name"])
            if "mpmpc" in k:
                out.setdefault(wl, {}).setdefault(k, {})["_trace"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                                     "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
# derived shares
for wl, ks in out.items():
    if not isinstance(ks, dict):
        continue
    for k, c in ks.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if not wc:
            continue
        d = {}
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
                     "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_VALU2"):
            if name in c:
                d[name + "/WAVE_CYCLES"] = c[name] / wc
        w = c.get("SQ_WAVES")
        if w:
            for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_IFETCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                         "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32"):
                if name in c:
                    d[name + "/wave"] = c[name] / w
            d["wave_cycles/wave (quad)"] = wc / w
        if c.get("SQC_ICACHE_REQ"):
            d["icache_hit_rate"] = c.get("SQC_ICACHE_HITS", 0.0) / c["SQC_ICACHE_REQ"]
            d["icache_miss_rate"] = c.get("SQC_ICACHE_MISSES", 0.0) / c["SQC_ICACHE_REQ"]
        c["_derived"] = d
path = os.path.join(P, "pmc_wait_%s.json" % tag)
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(path)
for wl, ks in out.items():
    if isinstance(ks, dict):
        for k, c in ks.items():
            print(wl, k, json.dumps(c.get("_derived", {}), indent=None), c.get("_trace"))
