for r in 1 2; do for q in 4 8; do for f in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 150 --in-flight $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hw queues $q in flight $f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3))"
done; done; done
