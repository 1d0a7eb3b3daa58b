#!/bin/bash
# VERDICT r5 missing 4: throughput + rocprofv3 evidence for the kernels round 5 added (non-diagonal weights, horizons above 63).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect_variants.sh r6v'
# Raw output: gpurun_out/<tag>/; python profiles/summarize_variants.py <tag> r6  ->  profiles/r6/variants.{json,md}
set -u
TAG=${1:-r6v}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
G_SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
G_MIX="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"
run() {
  name=$1; shift
  python3 "$R/profiles/variants.py" "$@" > "$O/$name.json" 2> "$O/$name.err"
  python3 "$R/profiles/variants.py" "$@" --pipeline 4 > "$O/${name}_p4.json" 2>> "$O/$name.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_trace" -- python3 "$R/profiles/variants.py" "$@" > "$O/${name}_trace.json" 2>> "$O/$name.err"
  rocprofv3 --pmc $G_SQ --output-format csv -d "$O/${name}_sq" -- python3 "$R/profiles/variants.py" "$@" --steps 2 --repeats 1 > "$O/${name}_sq.json" 2>> "$O/$name.err"
  rocprofv3 --pmc $G_MIX --output-format csv -d "$O/${name}_mix" -- python3 "$R/profiles/variants.py" "$@" --steps 2 --repeats 1 > "$O/${name}_mix.json" 2>> "$O/$name.err"
}
run full_N30 --weights full --N 30 --B 8192 --cfgid 4
run full_N30_free --weights full --N 30 --B 8192 --cfgid 2
for n in 64 127 128 255; do run stock_N$n --weights stock --N $n --B 8192 --cfgid 2; done
run stock_obst_N100 --weights stock --N 100 --B 8192 --cfgid 4
for n in 70 150; do
  run topt_N$n --weights time_optimal --N $n --B 2048 --cfgid 2
  run full_N$n --weights full --N $n --B 2048 --cfgid 4
  run bounded_N$n --weights bounded --N $n --B 2048 --cfgid 4
done
ls "$O" | wc -l
for f in "$O"/*.err; do [ -s "$f" ] && { echo "== $f"; tail -3 "$f"; }; done 2>/dev/null | head -40
cat "$O"/full_N30.json "$O"/stock_N64.json "$O"/stock_N255.json 2>/dev/null | cut -c1-400
