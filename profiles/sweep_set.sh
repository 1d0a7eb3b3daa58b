#!/bin/bash
# sweep one solver setting on the GPU over configs 2, 3, 4 and the large batch:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/sweep_set.sh ipm_tol 1e-9 1e-8 1e-7'
K=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  for c in 2 3 4; do
    python bench.py --no-cpu --config $c --steps 10 --warmup 2 --set $K=$v 2>/dev/null > gpurun_out/ss.json
    python - "$K" "$v" "$c" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ss.json"))
print("%s=%s cfg%s  %.3f M/s  %.4f ms  ipm mean %.2f max %d  status %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"],
      d["iters"]["ipm_mean"], d["iters"]["ipm_max"], d["status_counts"]))
PY
  done
  python bench.py --no-cpu --batch 65536 --steps 5 --warmup 1 --set $K=$v 2>/dev/null > gpurun_out/ss.json
  python -c "
import json; d=json.load(open('gpurun_out/ss.json')); print('$K=$v big  %.3f M/s  %.4f ms  ipm mean %.2f max %d' % (d['value']/1e6, d['ms_per_step'], d['iters']['ipm_mean'], d['iters']['ipm_max']))"
done
