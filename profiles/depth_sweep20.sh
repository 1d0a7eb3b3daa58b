#!/bin/bash
# launches in flight x hardware queues of the HIP runtime, 20- and 200-step timed regions (config 2) -> profiles/<round>/depth_sweep.txt
for q in 8 16; do for p in 3 4 5 6 8; do for s in 20 200; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --pipeline $p --steps $s --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q pipeline $p steps $s: %.2f M  ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"
done; done; done
