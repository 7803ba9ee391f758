# dev: bench.py with and without hip.bind_host_thread, the process started on each NUMA node in turn and unpinned (results -> gpurun_out/bind_ab.txt)
mkdir -p gpurun_out; O=gpurun_out/bind_ab.txt; : > $O
C0=$(cat /sys/devices/system/node/node0/cpulist); C1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null || echo $C0)
run() {  # label, taskset list or "", extra flags
  if [ -n "$2" ]; then PRE="taskset -c $2"; else PRE=""; fi
  $PRE python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; h=d['host']
print('$1', 'value', round(d['value'],1), 'seq', round(d['sequential']['value'],1), 'gemm3', k['gemm_kernel<3>'], 'frac', round(d['roofline']['frac'],4), 'pipe', round(d['roofline']['matrix_pipe']['frac'],4), 'bound', h['submitting_thread_bound_to_gpu_numa_node'], 'gpu node', h['gpu_numa_node'], 'cpus', h['cpus_during_gpu_legs'])" >> $O
}
for r in 1 2; do
  run "started on node 0, bind   " "$C0" ""
  run "started on node 0, no bind" "$C0" "--no-host-bind"
  run "started on node 1, bind   " "$C1" ""
  run "started on node 1, no bind" "$C1" "--no-host-bind"
  run "unpinned, bind            " "" ""
  run "unpinned, no bind         " "" "--no-host-bind"
done
cat $O
