#!/bin/bash
# everything profiles/r4 is made of, on one box, for the library in the tree:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect_all.sh'
# then here:  python profiles/summarize.py r4; python profiles/summarize_wait.py r4w r4; python profiles/kernel_resources.py r4
#             (python profiles/finish_collect.py does all of it)
bash profiles/collect.sh r4 > /dev/null 2>&1
bash profiles/collect_wait.sh r4w > /dev/null 2>&1
bash profiles/phases.sh r4 > /dev/null 2>&1
bash profiles/phases_tt.sh r4 > /dev/null 2>&1
python profiles/rollout_warm.py > gpurun_out/rollout_warm.txt 2>&1
MPMPC_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --single-process --config 5 --steps 20 > gpurun_out/single_process_2handles.json 2> gpurun_out/single_process_2handles.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/torchrun_1rank.json 2> gpurun_out/torchrun_1rank.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ls gpurun_out/r4 | wc -l; ls gpurun_out/r4w | wc -l; tail -c 300 gpurun_out/torchrun_1rank.json
