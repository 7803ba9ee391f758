for cfg in "8 4" "8 0" "4 4"; do set -- $cfg
GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu-baseline --no-roi-load --in-flight $2 --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $1 in-flight $2: value', round(d['value'],1), 'fp32 pipe', round(d['fp32_mfma_pipe']['value'],1), d['fp32_mfma_pipe']['achieved_tflops_dominant_kernel'])"
done
