#!/bin/bash
# A/B of library variants in profiles/_ab/<name>.so on one box: bash profiles/ab_libs.sh A R2 R3
D=multi-purpose-mpc_amd/csrc
cp $D/libmpmpc.so /tmp/keep.so
for v in "$@"; do
  cp profiles/_ab/$v.so $D/libmpmpc.so
  for a in "--steps 200" "--config 4 --steps 30 --warmup 3" "--config 5 --steps 30 --warmup 3" "--batch 65536 --steps 8 --warmup 2" "--batch 4096 --steps 50 --warmup 5" "--batch 2048 --steps 50 --warmup 5" "--batch 16384 --steps 20 --warmup 3"; do
    python bench.py --no-cpu $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$a', '%.2f M/s' % (d['value']/1e6), 'ipm %.2f/%d' % (d['iters']['ipm_mean'], d['iters']['ipm_max']), d['status_counts'])"
  done
done
cp /tmp/keep.so $D/libmpmpc.so
