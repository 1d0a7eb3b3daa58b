#!/bin/bash
# Reduced-native kernels against the general ones, and one against two waves per SIMD (MPMPC_RN_OCC), on one box.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/rn_ab.sh tag'
T=${1:-rn}
O=gpurun_out/$T
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || tail -3 $O/$name.err; }
for rep in 1 2; do
  run cfg2_gen_$rep python bench.py --no-cpu --steps 200 --set native=0
  run cfg2_rn2_$rep python bench.py --no-cpu --steps 200
  MPMPC_RN_OCC=1 run cfg2_rn1_$rep python bench.py --no-cpu --steps 200
done
for b in 2048 4096; do
  run b${b}_g64_gen python bench.py --no-cpu --steps 50 --batch $b --lanes 64 --set native=0
  run b${b}_g32_gen python bench.py --no-cpu --steps 50 --batch $b --lanes 32 --set native=0
  run b${b}_g64_rn2 python bench.py --no-cpu --steps 50 --batch $b --lanes 64
  MPMPC_RN_OCC=1 run b${b}_g64_rn1 python bench.py --no-cpu --steps 50 --batch $b --lanes 64
  run b${b}_g32_rn2 python bench.py --no-cpu --steps 50 --batch $b --lanes 32
  MPMPC_RN_OCC=1 run b${b}_g32_rn1 python bench.py --no-cpu --steps 50 --batch $b --lanes 32
done
run cfg4_gen python bench.py --no-cpu --config 4 --steps 20 --set native=0
run cfg4_rn2 python bench.py --no-cpu --config 4 --steps 20
MPMPC_RN_OCC=1 run cfg4_rn1 python bench.py --no-cpu --config 4 --steps 20
run cfg4_rn2_g64 python bench.py --no-cpu --config 4 --steps 20 --lanes 64
run big_gen python bench.py --no-cpu --batch 65536 --steps 5 --warmup 1 --set native=0
run big_rn2 python bench.py --no-cpu --batch 65536 --steps 5 --warmup 1
MPMPC_RN_OCC=1 run big_rn1 python bench.py --no-cpu --batch 65536 --steps 5 --warmup 1
run big_rn2_g64 python bench.py --no-cpu --batch 65536 --steps 5 --warmup 1 --lanes 64
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print("%-18s %12.0f solves/s  %8.4f ms/step  k2 %.4f ms  ipm %.2f/%d  %s" % (os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["roofline"]["avg_ms"],
              d["iters"]["ipm_mean"], d["iters"]["ipm_max"], d["status_counts"]))
    except Exception as e:
        print(os.path.basename(f), "no line:", e)
PY
