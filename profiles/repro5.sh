#!/bin/bash
# is config 5's pipelined rate erratic?  five runs each of configs 5 and 4 on one box
for i in 1 2 3 4 5; do for c in 5 4; do
python bench.py --no-cpu --config $c --steps 20 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run $i config $c: %.2f M  ms/step median %.4f min %.4f max %.4f  one-in-flight %.2f M' % (d['value']/1e6, d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['value_one_launch_in_flight']/1e6))"
done; done
