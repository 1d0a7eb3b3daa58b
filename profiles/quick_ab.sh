#!/bin/bash
# same-box check used while trimming the kernels: GPU parity tests, then the bench lines that matter
mkdir -p gpurun_out/e2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/e2/pytest.log
for a in "--steps 200" "--steps 20 --warmup 5" "--config 4 --steps 30 --warmup 3" "--batch 65536 --steps 8 --warmup 2" "--config 3 --steps 20 --warmup 3"; do
  python bench.py --no-cpu $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$a', '%.2f M/s' % (d['value']/1e6), 'one %.2f' % ((d.get('value_one_launch_in_flight') or 0)/1e6), 'ipm %.2f/%d' % (d['iters']['ipm_mean'], d['iters']['ipm_max']), d['status_counts'], d.get('max_abs_u_minus_uref'))"
done
