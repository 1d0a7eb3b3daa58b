#!/bin/bash
# quick GPU check used while tuning: parity tests, then the single-GPU configurations and the large batch
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/quick.sh tag'
T=${1:-q}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/${T}_pytest.log
for c in 2 3 4 5; do python bench.py --config $c --steps 10 --warmup 2 > gpurun_out/${T}_$c.json 2> gpurun_out/${T}_$c.err || tail -5 gpurun_out/${T}_$c.err; done
python bench.py --no-cpu --batch 65536 --steps 5 --warmup 1 > gpurun_out/${T}_big.json 2>/dev/null
python - "$T" <<'PY'
import json, sys
for c in ("2", "3", "4", "5", "big"):
    try:
        d = json.load(open("gpurun_out/%s_%s.json" % (sys.argv[1], c)))
    except Exception as e:
        print(c, "no line:", e)
        continue
    print(c, round(d["value"]), round(d["ms_per_step"], 4), round(d["roofline"]["avg_ms"], 4), d["iters"], d["status_counts"],
          d.get("max_abs_u_minus_uref"), d.get("status_agreement"))
PY
