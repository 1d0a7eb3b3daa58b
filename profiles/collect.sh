#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect.sh r4'
# Raw output lands in gpurun_out/<round>/ (scratch); profiles/summarize.py turns it into the
# committed summaries under profiles/<round>/.
set -u
ROUND=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$ROUND
mkdir -p "$O"
cd "$R"
python bench.py > "$O/bench_cfg2.json" 2> "$O/bench_cfg2.err"
for c in 3 4 5; do python bench.py --config $c --steps 20 --warmup 2 > "$O/bench_cfg$c.json" 2>/dev/null; done
# the verdict semantics ADVICE r2 asked to see side by side: every proven infeasibility reported (phase1_accept = 0)
for c in 4 5; do python bench.py --config $c --steps 20 --warmup 2 --no-cpu --set phase1_accept=0 > "$O/bench_cfg${c}_strict.json" 2>/dev/null; done
python bench.py --batch 65536 --steps 5 --warmup 1 --no-cpu > "$O/bench_cfg2_b65536.json" 2>/dev/null
cd /tmp; export TMPDIR=/tmp
# (counter passes serialise the kernels and are slow per launch: few repeats, short clock ramp)
PMC="--repeats 2 --prewarm 30"
prof() { out=$1; shift; rocprofv3 "$@" --output-format csv -d "$O/$out" -- python3 "$R/bench.py" ${BENCH_ARGS:-} --no-cpu --no-extra-legs > "$O/$out.json" 2>/dev/null; }
BENCH_ARGS="" prof trace --kernel-trace --stats
# ... and of the driver's own command (20-step regions)
BENCH_ARGS="--gpus 1 --steps 20 --warmup 5" prof trace_driver --kernel-trace --stats
BENCH_ARGS="--steps 5 --warmup 1 $PMC" prof pmc_fetch --pmc FETCH_SIZE
BENCH_ARGS="--steps 5 --warmup 1 $PMC" prof pmc_write --pmc WRITE_SIZE
BENCH_ARGS="--config 4 --steps 5 --warmup 1 $PMC" prof pmc_sq1_cfg4 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
BENCH_ARGS="--config 4 --steps 5 --warmup 1 $PMC" prof pmc_sq2_cfg4 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT
BENCH_ARGS="--batch 65536 --steps 3 --warmup 1 $PMC" prof pmc_sq1_b65536 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
BENCH_ARGS="--batch 65536 --steps 3 --warmup 1 $PMC" prof pmc_sq2_b65536 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT
BENCH_ARGS="--steps 5 --warmup 1 $PMC" prof pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
BENCH_ARGS="--steps 5 --warmup 1 $PMC" prof pmc_sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT
BENCH_ARGS="--config 3 --steps 20 --warmup 2" prof trace_cfg3 --kernel-trace --stats
BENCH_ARGS="--config 4 --steps 20 --warmup 2" prof trace_cfg4 --kernel-trace --stats
BENCH_ARGS="--config 5 --steps 20 --warmup 2" prof trace_cfg5 --kernel-trace --stats
BENCH_ARGS="--batch 65536 --steps 5 --warmup 1" prof trace_b65536 --kernel-trace --stats
BENCH_ARGS="--batch 65536 --steps 3 --warmup 1 $PMC" prof pmc_fetch_b65536 --pmc FETCH_SIZE
BENCH_ARGS="--batch 65536 --steps 3 --warmup 1 $PMC" prof pmc_write_b65536 --pmc WRITE_SIZE
# HBM traffic of the other BASELINE configurations (one launch of config 3 / 4 / 5 = the reduced-native kernel + its tail kernel)
for c in 3 4 5; do
  BENCH_ARGS="--config $c --steps 5 --warmup 1 $PMC" prof pmc_fetch_cfg$c --pmc FETCH_SIZE
  BENCH_ARGS="--config $c --steps 5 --warmup 1 $PMC" prof pmc_write_cfg$c --pmc WRITE_SIZE
done
lscpu | grep -E "Model name|^CPU\(s\)" > "$O/host.txt"
ls "$O"
