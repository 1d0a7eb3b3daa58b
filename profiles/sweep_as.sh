#!/bin/bash
# active-set rounds per attempt (as_rounds) and the add fraction on the BASELINE configurations
for kv in "as_rounds=4" "as_rounds=3" "as_rounds=2" "as_rounds=1" "as_add_fraction=0.5" "as_add_fraction=0.1"; do
  for a in "--config 3 --steps 20 --warmup 3" "--steps 200" "--config 4 --steps 30 --warmup 3" "--batch 65536 --steps 8 --warmup 2"; do
    python bench.py --no-cpu $a --set $kv 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$kv', '$a', '%.2f M/s' % (d['value']/1e6), 'ipm %.2f/%d' % (d['iters']['ipm_mean'], d['iters']['ipm_max']), d['status_counts'])"
  done
done
