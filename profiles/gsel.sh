for B in 1536 2048 3072 4096 8192; do
  for g in 64 32 16; do
    python bench.py --lanes $g --config 2 --batch $B --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('B', $B, 'G', $g, round(d['value']), 'ms', round(d['ms_per_step'],4))"
  done
done
