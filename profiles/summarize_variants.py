"""Condense gpurun_out/<tag>/ of profiles/collect_variants.sh into profiles/<round>/variants.json and variants.md: per variant the
HIP-timed rates (one / four launches in flight), per kernel the rocprofv3 kernel-trace average, its resources as the profiler
read them off the dispatch (VGPR, accumulation VGPR, LDS, scratch), and the SQ counters per launch and per instance.

    python profiles/summarize_variants.py r6v r6
"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r6v"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r6"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", rnd)
os.makedirs(P, exist_ok=True)


def short(k):
    m = re.match(r"(?:void )?(mpmpc_\w+(?:<[^>]*>)?)", k)
    return m.group(1) if m else k


def load(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


sys.path.insert(0, os.path.join(root, "profiles"))
import kernel_resources  # noqa: E402
CO = {re.sub(r"\s+", " ", r["name"]).replace("mpmpc::", ""): {"vgpr": r["vgpr"], "agpr": r["agpr"], "scratch": r["scratch"], "lds_static": r["lds"]}
      for r in kernel_resources.kernel_table()}
# dynamic LDS of the workgroup kernels (lane_gpu.hpp: LaneBlock::lds_bytes; mpmpc_hip.hip: RNB_SLOTS = 40, general 66 slots)
def dyn_lds(k):
    if k.startswith("mpmpc_reduced_pair_block_kernel"):          # LaneBlock<128, 74, 128, 4>: 37 pair slots + 4 exchange rows
        return 8 * ((74 + 4) * 128 + 8 + 8)
    if k.startswith(("mpmpc_reduced_t_pair_block_kernel", "mpmpc_reduced_tail_pair_block_kernel")):      # <128, 80, 128, 4>
        return 8 * ((80 + 4) * 128 + 8 + 8)
    m = re.match(r"mpmpc_(reduced|solve)_block_kernel<(\d+)", k)
    if not m:
        return 0
    G = int(m.group(2)); slots = 40 if m.group(1) == "reduced" else 66
    return 8 * ((slots + 9) * G + 8 + 18)


out = {}
names = sorted(os.path.basename(f)[:-5] for f in glob.glob(os.path.join(O, "*.json"))
               if not re.search(r"_(p4|trace|sq|mix)\.json$", f))
for name in names:
    one, p4 = load(os.path.join(O, name + ".json")), load(os.path.join(O, name + "_p4.json"))
    if not one:
        continue
    v = {k: one[k] for k in ("library", "weights", "N", "B", "cfgid", "lanes_per_instance", "ipm_iters_mean", "ipm_iters_max", "admm_iters_mean", "status_counts")}
    v["solves_per_s_one_launch_in_flight"] = one["solves_per_s"]
    v["ms_per_step_one_launch_in_flight"] = one["ms_per_step"]
    if p4:
        v["solves_per_s_four_launches_in_flight"] = p4["solves_per_s"]
    kern = {}
    for f in glob.glob(os.path.join(O, name + "_trace", "*", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "mpmpc" in r["Name"]:
                kern.setdefault(short(r["Name"]), {})["trace"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                                "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                                                "share_pct": float(r["Percentage"])}
    for grp in ("sq", "mix"):
        for f in glob.glob(os.path.join(O, name + "_" + grp, "*", "*counter_collection.csv")):
            agg = collections.defaultdict(list)
            res = {}
            for r in csv.DictReader(open(f)):
                if "mpmpc" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                res[k] = {"vgpr": int(r["VGPR_Count"]), "accum_vgpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                          "lds_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]), "workgroup": int(r["Workgroup_Size"]),
                          "grid": int(r["Grid_Size"])}
            for (k, c), vals in agg.items():
                kern.setdefault(k, {}).setdefault("pmc_per_launch", {})[c] = sum(vals) / len(vals)
            for k, rr in res.items():
                kern.setdefault(k, {})["resources"] = rr
    for k, d in kern.items():
        c = d.get("pmc_per_launch", {})
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
            d["valu_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            wg = d.get("resources", {}).get("workgroup", 64)
            inst_per_launch = c["SQ_WAVES"] * 64 / max(wg, 64) if wg > 64 else None      # workgroup kernels: one instance per workgroup
            if inst_per_launch is None and (k.startswith("mpmpc_reduced_pair_kernel<64>") or k.startswith("mpmpc_reduced_t_pair_kernel<64>")):
                inst_per_launch = c["SQ_WAVES"]          # one instance per wavefront
            if inst_per_launch is None and k.startswith("mpmpc_reduced_pair_kernel<16>"):
                inst_per_launch = 4 * c["SQ_WAVES"]
            if inst_per_launch:
                d["valu_per_instance"] = c["SQ_INSTS_VALU"] / inst_per_launch
            elif k.startswith("mpmpc_solve_kernel<64"):          # the general wavefront kernels: one instance per wave
                d["valu_per_instance"] = d["valu_per_wave"]
            f64 = sum(c.get(n, 0.0) for n in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
            if f64:
                d["fp64_fraction_of_valu"] = f64 / c["SQ_INSTS_VALU"]
            if c.get("SQ_WAVE_CYCLES"):
                for n in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
                    if n in c:
                        d[n.lower() + "_share_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    for k, d in kern.items():          # registers / scratch / static LDS from the code object the run loaded (profiles/kernel_resources.py)
        if k in CO:
            d["code_object"] = CO[k]
    v["kernels"] = kern
    out[name] = v
json.dump(out, open(os.path.join(P, "variants.json"), "w"), indent=1, sort_keys=True)

lines = ["# Kernel variants beside the BASELINE configurations, measured (profiles/collect_variants.sh -> profiles/summarize_variants.py)", "",
         "Library: %s.  One resident batch, HIP-timed regions of 20 launches (median of 5); per kernel: rocprofv3 `--kernel-trace --stats`" % next(iter(out.values()))["library"],
         "average of the same command, registers / scratch / static LDS from the code object (profiles/kernel_resources.py), SQ counters from two `--pmc` passes.", "",
         "| variant | N | B | lanes / instance | solves/s, 1 launch in flight | 4 in flight | interior-point iterations mean (max) | statuses |",
         "|---|---|---|---|---|---|---|---|"]
for name, v in out.items():
    lines.append("| %s | %d | %d | %d | %.2f M | %s | %.2f (%d) | %s |" % (
        name, v["N"], v["B"], v["lanes_per_instance"], v["solves_per_s_one_launch_in_flight"] / 1e6,
        ("%.2f M" % (v["solves_per_s_four_launches_in_flight"] / 1e6)) if "solves_per_s_four_launches_in_flight" in v else "-",
        v["ipm_iters_mean"], v["ipm_iters_max"], " ".join("%s:%d" % kv for kv in sorted(v["status_counts"].items()))))
lines += ["", "| variant | kernel | calls | trace avg us | registers (of which AGPR; code object) | LDS KB (static + dynamic) | scratch B | VALU / wave | VALU / instance | FP64 share | ACTIVE_VALU | WAIT_ANY | WAIT_INST_LDS |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for name, v in out.items():
    for k, d in sorted(v["kernels"].items()):
        t, r = d.get("trace", {}), d.get("code_object", {})

        def pct(key):
            return "%.0f %%" % (100 * d[key]) if key in d else "-"
        lines.append("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            name, k, t.get("calls", "-"), ("%.1f" % t["avg_us"]) if t else "-",
            ("%d (%d)" % (r["vgpr"], r["agpr"])) if r else "-", ("%.1f" % ((r["lds_static"] + dyn_lds(k)) / 1024.0)) if r else "-",
            r.get("scratch", "-"), ("%.0f" % d["valu_per_wave"]) if "valu_per_wave" in d else "-",
            ("%.0f" % d["valu_per_instance"]) if "valu_per_instance" in d else "-", pct("fp64_fraction_of_valu"),
            pct("sq_active_inst_valu_share_of_wave_cycles"), pct("sq_wait_any_share_of_wave_cycles"), pct("sq_wait_inst_lds_share_of_wave_cycles")))
open(os.path.join(P, "variants.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
