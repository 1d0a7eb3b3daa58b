#!/bin/bash
# configs 4 / 5, default and strict (phase1_accept = 0): whole-step rate and the per-kernel split of one profiled run
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/tail_ab.sh tag'
T=${1:-tail}
O=gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in 4 5; do
  python bench.py --no-cpu --config $c --steps 20 --warmup 3 > $O/cfg${c}.json 2> $O/cfg${c}.err
  python bench.py --no-cpu --config $c --steps 20 --warmup 3 --set phase1_accept=0 > $O/cfg${c}s.json 2> $O/cfg${c}s.err
done
python bench.py --no-cpu --steps 200 > $O/cfg2.json 2> $O/cfg2.err
rocprofv3 --kernel-trace --stats -d $O/prof4 -o p4 -- python3 bench.py --no-cpu --config 4 --steps 20 --warmup 3 > /dev/null 2> $O/prof4.err
rocprofv3 --kernel-trace --stats -d $O/prof4s -o p4s -- python3 bench.py --no-cpu --config 4 --steps 20 --warmup 3 --set phase1_accept=0 > /dev/null 2> $O/prof4s.err
python - $O <<'PY'
import json, sys, glob, csv, os
O = sys.argv[1]
for f in sorted(glob.glob(O + "/cfg*.json")):
    try:
        d = json.load(open(f)); print("%-8s %12.0f solves/s %8.4f ms/step %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["status_counts"]))
    except Exception as e:
        print(f, "no line", e)
for p in ("prof4", "prof4s"):
    for f in glob.glob(O + "/%s/**/*kernel_stats.csv" % p, recursive=True):
        print(p)
        for r in list(csv.DictReader(open(f)))[:4]:
            print("   %-60s calls %5s avg %10.1f ns  total %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
        for g in glob.glob(os.path.dirname(f) + "/*"):
            if not g.endswith("kernel_stats.csv"): os.remove(g)
PY
