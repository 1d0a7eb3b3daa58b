timeout 900 python -m pytest tests/test_corridor.py tests/test_rollout.py -m gpu -x -q 2>&1 | tail -15
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nr -- python3 $R/profiles/next_rows.py > $R/gpurun_out/next_rows.txt 2>&1
cat $R/gpurun_out/next_rows.txt | tail -8
find $R/gpurun_out/nr -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150
