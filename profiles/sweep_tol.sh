#!/bin/bash
# interior-point tolerance of the early attempt (when the active-set rounds take over), longer runs:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/sweep_tol.sh ipm_tol 1e-8 1e-7 2e-7 3e-7'
K=$1; shift
for v in "$@"; do
  for a in "--steps 200" "--config 4 --steps 30 --warmup 3" "--config 5 --steps 30 --warmup 3" "--batch 65536 --steps 8 --warmup 2" "--batch 4096 --steps 50 --warmup 5"; do
    python bench.py --no-cpu $a --set $K=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$K=$v', '$a', '%.2f M/s' % (d['value']/1e6), 'ipm %.2f/%d' % (d['iters']['ipm_mean'], d['iters']['ipm_max']), d['status_counts'])"
  done
done
