# dev: HIP hardware-queue count against batches in flight
for q in 4 8 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 120 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$q:', round(d['value'],1), '| sequential', round(d['sequential']['value'],1))"
done
