#!/bin/bash
# SQ counters of the two-stages-per-lane kernel against the one-stage kernel on the same batch:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/pmc_s2.sh s2pmc'
set -u
TAG=${1:-s2pmc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
G_SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
G_MIX="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"
G_X="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_VALU_INT32"
for cfg in "2 65536" "2 1024" "4 8192"; do
  set -- $cfg
  for lanes in 32 16; do
    name=cfg$1_B$2_l$lanes
    A="--weights stock --N 30 --B $2 --cfgid $1 --lanes $lanes --pipeline 4"
    python3 "$R/profiles/variants.py" $A > "$O/$name.json" 2> "$O/$name.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_trace" -- python3 "$R/profiles/variants.py" $A > "$O/${name}_trace.json" 2>> "$O/$name.err"
    rocprofv3 --pmc $G_SQ --output-format csv -d "$O/${name}_sq" -- python3 "$R/profiles/variants.py" $A --steps 2 --repeats 1 > "$O/${name}_sq.json" 2>> "$O/$name.err"
    rocprofv3 --pmc $G_MIX --output-format csv -d "$O/${name}_mix" -- python3 "$R/profiles/variants.py" $A --steps 2 --repeats 1 > "$O/${name}_mix.json" 2>> "$O/$name.err"
    rocprofv3 --pmc $G_X --output-format csv -d "$O/${name}_x" -- python3 "$R/profiles/variants.py" $A --steps 2 --repeats 1 > "$O/${name}_x.json" 2>> "$O/$name.err"
  done
done
cat "$O"/*l16.json "$O"/*l32.json | cut -c1-300
