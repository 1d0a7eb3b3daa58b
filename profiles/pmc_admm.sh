#!/bin/bash
# PMC cost of one ADMM iteration of K2: two pure-ADMM runs (100 and 300 iterations, no checks, no polish),
# counters differenced by profiles/summarize-style post-processing.  /usr/local/graft/bin/gpurun -- 'bash profiles/pmc_admm.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_admm
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for it in 100 300; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d "$O/it$it" -- python3 "$R/bench.py" --no-cpu --steps 3 --warmup 1 \
    --set polish=0 --set check_termination=0 --set adaptive_rho=0 --set max_iter=$it > /dev/null 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
res = {}
for it in (100, 300):
    acc = collections.defaultdict(list)
    for f in glob.glob("%s/it%d/*/*counter_collection.csv" % (sys.argv[1], it)):
        for r in csv.DictReader(open(f)):
            if "solve" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[it] = {k: sum(v) / len(v) for k, v in acc.items()}
w = res[100].get("SQ_WAVES", 1024)
for k in sorted(res[100]):
    print("%-22s per wave per ADMM iteration: %10.1f" % (k, (res[300][k] - res[100][k]) / 200.0 / w))
PY
