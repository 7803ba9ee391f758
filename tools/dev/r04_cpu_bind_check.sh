# dev: does putting the GPU legs on one NUMA node change the CPU baseline that follows (it gets the original mask back)?
for f in "" "--no-host-bind" "" "--no-host-bind"; do
  timeout 400 python bench.py --steps 10 --no-roi-load --no-fp32-pipe $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cpu_baseline']
print('bind' if d['host']['submitting_thread_bound_to_gpu_numa_node'] else 'no bind', 'cpu_baseline', round(c['value'],3), 'cores', c['cores'], c['sample'][c['sample'].find('per batch'):][:60])" >> gpurun_out/cpu_bind_check.txt
done
cat gpurun_out/cpu_bind_check.txt
