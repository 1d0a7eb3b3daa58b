#!/bin/bash
# Round 3: where do the parked cycles (SQ_WAIT_ANY) of the solve kernels go?  Instruction-fetch / instruction-cache,
# LDS-issue and scalar-memory counters plus the instruction mix, for
#   cfg2   B = 1 024  one instance per wave   mpmpc_solve_kernel<64,16,false,2>
#   cfg4   B = 8 192  packed                  (+ its tail launch)
#   big    B = 65 536 packed
# One counter group per rocprofv3 pass (no trace domain beside --pmc); the program goes directly after `--`.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/collect_wait.sh r3a'
# profiles/summarize_wait.py condenses gpurun_out/<tag>/ into profiles/r3/pmc_wait_<tag>.json
set -u
TAG=${1:-r6w}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
prof() { out=$1; shift; rocprofv3 "$@" --output-format csv -d "$O/$out" -- python3 "$R/bench.py" ${BENCH_ARGS:-} --no-cpu --no-extra-legs > "$O/$out.json" 2>"$O/$out.err"; }
G_WAIT="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"
G_FETCH="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
G_ICACHE="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
G_ICACHE2="SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_DCACHE_REQ SQ_INSTS_BRANCH"
G_MIX="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU"
G_MEM="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
for wl in cfg2 cfg3 cfg4 big; do
  case $wl in
    cfg2) A="--steps 5 --warmup 1 --repeats 2 --prewarm 30" ;;
    cfg3) A="--config 3 --steps 5 --warmup 1 --repeats 2 --prewarm 30" ;;
    cfg4) A="--config 4 --steps 5 --warmup 1 --repeats 2 --prewarm 30" ;;
    big)  A="--batch 65536 --steps 3 --warmup 1 --repeats 2 --prewarm 30" ;;
  esac
  BENCH_ARGS="$A" prof ${wl}_wait --pmc $G_WAIT
  BENCH_ARGS="$A" prof ${wl}_fetch --pmc $G_FETCH
  BENCH_ARGS="$A" prof ${wl}_icache --pmc $G_ICACHE
  BENCH_ARGS="$A" prof ${wl}_icache2 --pmc $G_ICACHE2
  BENCH_ARGS="$A" prof ${wl}_mix --pmc $G_MIX
  BENCH_ARGS="$A" prof ${wl}_mem --pmc $G_MEM
  BENCH_ARGS="${A/--steps 5/--steps 20}" prof ${wl}_trace --kernel-trace --stats
done
ls "$O" | head -50
for f in "$O"/*.err; do [ -s "$f" ] && { echo "== $f"; tail -3 "$f"; }; done 2>/dev/null | head -60
