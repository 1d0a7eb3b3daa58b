#!/bin/bash
# the second half of profiles/collect_all.sh as a call of its own (the raw rocprofv3 output of both halves together exceeds what
# one gpurun call copies back):  /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash profiles/collect_rest.sh r6'
R=${1:-r6}
bash profiles/collect_wait.sh ${R}w > /dev/null 2>&1
bash profiles/phases.sh $R > /dev/null 2>&1
bash profiles/phases_tt.sh $R > /dev/null 2>&1
python profiles/rollout_warm.py > gpurun_out/rollout_warm.txt 2>&1
python profiles/branch_agreement.py > gpurun_out/branch_agreement.txt 2>&1
python profiles/long_horizon_timing.py > gpurun_out/long_horizon.txt 2>&1
bash profiles/depth_sweep20.sh > gpurun_out/depth_sweep.txt 2>&1
MPMPC_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --single-process --config 5 --steps 20 > gpurun_out/single_process_2handles.json 2> gpurun_out/single_process_2handles.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/torchrun_1rank.json 2> gpurun_out/torchrun_1rank.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
du -sh gpurun_out; ls gpurun_out/${R}w | wc -l; tail -c 300 gpurun_out/torchrun_1rank.json
