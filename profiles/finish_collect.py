"""After `gpurun -- 'bash profiles/collect_all.sh'`: condense gpurun_out/ into profiles/r4/ (run here, in the authoring container)."""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
for cmd in (["profiles/summarize.py", "r4"], ["profiles/summarize_wait.py", "r4w", "r4"], ["profiles/kernel_resources.py", "r4"]):
    subprocess.run([sys.executable] + cmd, check=True, stdout=subprocess.DEVNULL)
for f in glob.glob("gpurun_out/r4/phases_*.txt") + ["gpurun_out/rollout_warm.txt", "gpurun_out/torchrun_1rank.json"]:
    shutil.copy(f, "profiles/r4/")
src = json.load(open("profiles/r4/pmc_summary.json"))["library_source_hash"]
if os.path.exists("gpurun_out/single_process_2handles.json"):
    shutil.copy("gpurun_out/single_process_2handles.json", "profiles/r4/")
print("library", src, "tree", open("multi-purpose-mpc_amd/csrc/libmpmpc.srchash").read().strip())
for f in sorted(glob.glob("profiles/r4/bench_*.json")):
    d = json.load(open(f))
    print("%-28s %6.2f M  %.4f ms  kernel %.4f ms  %s  ipm %.2f" % (os.path.basename(f), d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_ms"],
                                                                 d.get("status_counts"), d["iters"]["ipm_mean"]))
w = json.load(open("profiles/r4/pmc_wait_r4w.json"))
for wl in ("cfg2", "cfg4", "big"):
    for k, v in w[wl].items():
        if "reduced" in k:
            d = v["_derived"]
            print(wl, k, "WAIT_ANY %.3f WAIT_INST %.3f VALU %.3f valu/wave %.0f" % (d["SQ_WAIT_ANY/WAVE_CYCLES"], d["SQ_WAIT_INST_ANY/WAVE_CYCLES"],
                  d["SQ_ACTIVE_INST_VALU/WAVE_CYCLES"], d["SQ_INSTS_VALU/wave"]), "avg_us %.1f" % v["_trace"]["avg_us"])
