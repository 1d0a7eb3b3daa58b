#!/bin/bash
# several builds (profiles/_ab/<name>.so ...) on the same box:  bash profiles/abn.sh "2 3" F0 F1 F2
CS=$1; shift
D=multi-purpose-mpc_amd/csrc
cp $D/libmpmpc.so /tmp/keep.so
for rep in 1 2; do
for v in "$@"; do
  [ -f profiles/_ab/$v.so ] || continue
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
done
cp /tmp/keep.so $D/libmpmpc.so
