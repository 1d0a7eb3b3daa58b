#!/bin/bash
# settings sweep on the device path (exploration): bash profiles/sweep.sh "<config list>" "KEY=V KEY=V ..." ...
#   each further argument is one variant (space-separated KEY=VALUE overrides, "" = defaults)
CS=$1; shift
for v in "$@"; do
  ARGS=""
  for kv in $v; do ARGS="$ARGS --set $kv"; done
  for c in $CS; do
    python bench.py --config $c --steps 10 --warmup 2 --no-cpu $ARGS > /tmp/sw.json 2>/dev/null
    python - "$v" "$c" <<'PY'
import json, sys
d = json.load(open("/tmp/sw.json"))
print("[%s] cfg %s: %d solves/s  K2 %.4f ms  ipm %.2f/%d  status %s" % (sys.argv[1], sys.argv[2], round(d["value"]),
      d["roofline"]["avg_ms"], d["iters"]["ipm_mean"], d["iters"]["ipm_max"], d["status_counts"]))
PY
  done
done
