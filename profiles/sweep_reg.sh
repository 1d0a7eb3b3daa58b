#!/bin/bash
# interior-point regularisation on the device (the emulation divides exactly; the device's reciprocals are 1-ulp approximations)
mkdir -p gpurun_out
for reg in 1e-8 5e-9 3e-9 2e-9; do
  for c in 2 4 3; do
    python bench.py --config $c --steps 30 --warmup 3 --set ipm_reg=$reg > gpurun_out/reg_${reg}_$c.json 2>/dev/null
    python - "$reg" "$c" <<'PY'
import json, sys
d = json.load(open("gpurun_out/reg_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("reg", sys.argv[1], "cfg", sys.argv[2], round(d["value"]), "ms", round(d["ms_per_step"], 4), "ipm", round(d["iters"]["ipm_mean"], 2), d["iters"]["ipm_max"], d["status_counts"], d.get("status_agreement"), d.get("max_abs_u_minus_uref"))
PY
  done
done
