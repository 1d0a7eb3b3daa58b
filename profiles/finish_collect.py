"""After `gpurun -- 'bash profiles/collect_all.sh'`: condense gpurun_out/ into profiles/<round>/ (run here, in the authoring container):
    python profiles/finish_collect.py [r5]"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r6"
os.chdir(ROOT)
for cmd in (["profiles/summarize.py", RND], ["profiles/summarize_wait.py", RND + "w", RND], ["profiles/kernel_resources.py", RND]):
    subprocess.run([sys.executable] + cmd, check=True, stdout=subprocess.DEVNULL)
for f in glob.glob("gpurun_out/%s/phases_*.txt" % RND) + ["gpurun_out/rollout_warm.txt", "gpurun_out/torchrun_1rank.json", "gpurun_out/branch_agreement.txt", "gpurun_out/long_horizon.txt", "gpurun_out/depth_sweep.txt"]:
    shutil.copy(f, "profiles/%s/" % RND)
src = json.load(open("profiles/%s/pmc_summary.json" % RND))["library_source_hash"]
if os.path.exists("gpurun_out/single_process_2handles.json"):
    shutil.copy("gpurun_out/single_process_2handles.json", "profiles/%s/" % RND)
print("library", src, "tree", open("multi-purpose-mpc_amd/csrc/libmpmpc.srchash").read().strip())
for f in sorted(glob.glob("profiles/%s/bench_*.json" % RND)):
    d = json.load(open(f))
    print("%-28s %6.2f M  %.4f ms  kernel %.4f ms  %s  ipm %.2f" % (os.path.basename(f), d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_ms"],
                                                                 d.get("status_counts"), d["iters"]["ipm_mean"]))
w = json.load(open("profiles/%s/pmc_wait_%sw.json" % (RND, RND)))
for wl in ("cfg2", "cfg4", "big"):
    for k, v in w[wl].items():
        if "reduced" in k:
            d = v["_derived"]
            print(wl, k, "WAIT_ANY %.3f WAIT_INST %.3f VALU %.3f valu/wave %.0f" % (d["SQ_WAIT_ANY/WAVE_CYCLES"], d["SQ_WAIT_INST_ANY/WAVE_CYCLES"],
                  d["SQ_ACTIVE_INST_VALU/WAVE_CYCLES"], d["SQ_INSTS_VALU/wave"]), "avg_us %.1f" % v["_trace"]["avg_us"])
