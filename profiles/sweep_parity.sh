#!/bin/bash
# like sweep.sh, with the parity columns (slower: runs the CPU oracle beside each variant)
CS=$1; shift
for v in "$@"; do
  ARGS=""
  for kv in $v; do ARGS="$ARGS --set $kv"; done
  for c in $CS; do
    python bench.py --config $c --steps 10 --warmup 2 $ARGS > /tmp/sw.json 2>/dev/null
    python - "$v" "$c" <<'PY'
import json, sys
d = json.load(open("/tmp/sw.json"))
print("[%s] cfg %s: %d solves/s  K2 %.4f ms  ipm %.2f/%d  status %s  |u-uref| %.2e  plan %.2e  agree %s" % (
      sys.argv[1], sys.argv[2], round(d["value"]), d["roofline"]["avg_ms"], d["iters"]["ipm_mean"], d["iters"]["ipm_max"],
      d["status_counts"], d["max_abs_u_minus_uref"], d["max_abs_plan_minus_ref"], d["status_agreement"]))
PY
  done
done
