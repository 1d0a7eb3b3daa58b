cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p4 -- python3 $R/bench.py --config 4 --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/p4.json 2>/dev/null
find $R/gpurun_out/p4 -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
