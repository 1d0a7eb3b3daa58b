#!/bin/bash
# A/B two builds of libmpmpc.so (profiles/_ab/{A,B}.so) on the same box:
#   /usr/local/graft/bin/gpurun -- 'bash profiles/ab.sh "2 3 4"'
CS=${1:-"2 4"}
D=multi-purpose-mpc_amd/csrc
cp $D/libmpmpc.so /tmp/keep.so
for v in A B A B; do
  cp profiles/_ab/$v.so $D/libmpmpc.so
  for c in $CS; do
    python bench.py --config $c --steps 10 --warmup 2 --no-cpu > /tmp/ab.json 2>/dev/null
    python - "$v" "$c" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(sys.argv[1], "cfg", sys.argv[2], round(d["value"]), "ms", round(d["roofline"]["avg_ms"], 4))
PY
  done
done
cp /tmp/keep.so $D/libmpmpc.so
