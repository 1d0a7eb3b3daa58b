#!/bin/bash
# profiles/micro/dep_wait (what SQ_WAIT_ANY is made of) timed and then measured with the SQ counters of collect_wait.sh
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash profiles/micro_wait.sh r3'
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/micro_$TAG
mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
"$R/profiles/micro/dep_wait" 2000 > "$O/dep_wait.txt" 2>&1
cat "$O/dep_wait.txt"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU \
  --output-format csv -d "$O/pmc" -- "$R/profiles/micro/dep_wait" 2000 > "$O/dep_wait_pmc.txt" 2>"$O/pmc.err"
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in sorted(acc.items()):
    m = {n: v[-1] for n, v in c.items()}            # last repetition
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        out[k] = {"wait_any": m["SQ_WAIT_ANY"] / wc, "wait_inst_any": m["SQ_WAIT_INST_ANY"] / wc, "active_valu": m["SQ_ACTIVE_INST_VALU"] / wc,
                  "active_any": m["SQ_ACTIVE_INST_ANY"] / wc, "valu_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
                  "salu_per_wave": m["SQ_INSTS_SALU"] / m["SQ_WAVES"], "quad_cycles_per_valu": wc / max(m["SQ_INSTS_VALU"], 1.0), "waves": m["SQ_WAVES"]}
json.dump(out, open(sys.argv[1] + "/dep_wait_pmc.json", "w"), indent=1)
for k, v in out.items():
    print("%-34s WAIT_ANY %.3f  WAIT_INST %.3f  ACTIVE_VALU %.3f  quad-cycles/VALU %.3f" % (k, v["wait_any"], v["wait_inst_any"], v["active_valu"], v["quad_cycles_per_valu"]))
PY
