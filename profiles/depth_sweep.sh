for q in 4 8; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q python profiles/launch_rate.py | grep "B  1024"
for c in 3 4; do for p in 2 3 4; do GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --config $c --pipeline $p --steps 50 --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c pipeline $p: %.2f M  ms/step %.4f' % (d['value']/1e6, d['ms_per_step']))"; done; done
for p in 2 3 4; do GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --batch 65536 --pipeline $p --steps 10 --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b65536 pipeline $p: %.2f M' % (d['value']/1e6))"; done
done
