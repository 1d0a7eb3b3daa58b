#!/bin/bash
# K1 (stand-alone assembly kernel) store variants and grid caps at HBM-resident sizes: roofline_assembly of bench.py
mkdir -p gpurun_out/k1
for v in 0 1 2 3; do for blocks in 2048 8192; do for B in 8192 65536; do
  # (the knobs MPMPC_K1_VARIANT / MPMPC_K1_BLOCKS existed in the library of this experiment only, commit "K1 variants")
  MPMPC_K1_VARIANT=$v MPMPC_K1_BLOCKS=$blocks python bench.py --config 2 --batch $B --no-cpu --repeats 3 --steps 5 --prewarm 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline_assembly']; print('variant $v blocks $blocks B $B: %.0f GB/s frac %.3f avg_ms %.4f' % (a['achieved'], a['frac'], a['avg_ms']))"
done; done; done | tee gpurun_out/k1/variants.txt
