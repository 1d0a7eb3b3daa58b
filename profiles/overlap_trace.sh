#!/bin/bash
# kernel trace of the two-batches-in-flight loop (bench.py --pipelined): do the tail launches of one handle run beside the
# packed launches of the other?   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/overlap_trace.sh 5'
C=${1:-5}
O=$PWD/gpurun_out/overlap
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/profiles/overlap_loop.py $C 40 > $O/line.txt 2> $O/err.txt
cd $R
python profiles/overlap_trace.py $O $C
