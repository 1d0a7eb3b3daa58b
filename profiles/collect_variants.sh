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
# round 5's additions as they run today ...
run full_N30 --weights full --N 30 --B 8192 --cfgid 4
run full_N30_free --weights full --N 30 --B 8192 --cfgid 2
for n in 70 150; do
  run full_N$n --weights full --N $n --B 2048 --cfgid 4
  run bounded_N$n --weights bounded --N $n --B 2048 --cfgid 4
done
# ... and the horizons above 63 of the reference's weights / of a terminal cost on t: round 6's two-stages-per-lane kernels (the
# launcher's choice) beside round 5's one-stage workgroup kernels (--lanes 128 / 256) on the same batches
for n in 64 127; do
  run stock_N${n} --weights stock --N $n --B 8192 --cfgid 2
  run stock_N${n}_r5 --weights stock --N $n --B 8192 --cfgid 2 --lanes 128
done
for n in 128 255; do
  run stock_N${n} --weights stock --N $n --B 8192 --cfgid 2
  run stock_N${n}_r5 --weights stock --N $n --B 8192 --cfgid 2 --lanes 256
done
run stock_obst_N100 --weights stock --N 100 --B 8192 --cfgid 4
run stock_obst_N100_r5 --weights stock --N 100 --B 8192 --cfgid 4 --lanes 128
run stock_obst_N150 --weights stock --N 150 --B 8192 --cfgid 4
run stock_obst_N150_r5 --weights stock --N 150 --B 8192 --cfgid 4 --lanes 256
run topt_N70 --weights time_optimal --N 70 --B 2048 --cfgid 3
run topt_N70_r5 --weights time_optimal --N 70 --B 2048 --cfgid 3 --lanes 128
run topt_N150 --weights time_optimal --N 150 --B 2048 --cfgid 3
run topt_N150_r5 --weights time_optimal --N 150 --B 2048 --cfgid 3 --lanes 256
# the pair layout at N = 30 (four instances per wavefront) beside the shipped packing
run stock_N30_pair16 --weights stock --N 30 --B 8192 --cfgid 4 --lanes 16
run stock_N30_r5 --weights stock --N 30 --B 8192 --cfgid 4 --lanes 32
ls "$O" | wc -l
for f in "$O"/*.err; do [ -s "$f" ] && { echo "== $f"; tail -3 "$f"; }; done 2>/dev/null | head -40
cat "$O"/full_N30.json "$O"/stock_N64.json "$O"/stock_N255.json 2>/dev/null | cut -c1-300
