#!/bin/bash
# A/B of the reduced-native tail kernel (same box): configs 4 and 5, knob MPMPC_LEAN_TAIL (the handle's initial mpmpc_set_tail_kernel mode)
mkdir -p gpurun_out/r4
for c in 4 5; do
  for k in 1 2 0 1 2 0; do   # 1 = tail kernel, two instances per wave (default); 2 = one per wave; 0 = general kernel
    echo "config $c lean $k: $(MPMPC_LEAN_TAIL=$k python bench.py --config $c --no-cpu 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
python bench.py --config 2 --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200
