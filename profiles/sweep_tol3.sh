#!/bin/bash
# the first attempt's interior-point tolerance on config 3 (terminal-time kernel)
for v in ${@:-1e-7 3e-7 1e-6 3e-6 1e-5 1e-4}; do
  python bench.py --no-cpu --config 3 --steps 20 --warmup 3 --set native_ipm_tol=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('native_ipm_tol=$v', '%.2f M/s' % (d['value']/1e6), 'one %.2f' % (d['value_one_launch_in_flight']/1e6), 'ipm %.2f/%d' % (d['iters']['ipm_mean'], d['iters']['ipm_max']), d['status_counts'], d.get('max_abs_u_minus_uref'))"
done
