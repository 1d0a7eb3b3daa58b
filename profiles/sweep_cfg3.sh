for v in 0.1 0.2 0.4 0.8 1.6; do
  for a in "--config 3 --steps 20 --warmup 3" "--steps 100"; do
    python bench.py --no-cpu $a --set ipm_start_dual=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ipm_start_dual=$v', '$a', '%.2f M/s' % (d['value']/1e6), 'ipm %.2f/%d' % (d['iters']['ipm_mean'], d['iters']['ipm_max']), d['status_counts'])"
  done
done
