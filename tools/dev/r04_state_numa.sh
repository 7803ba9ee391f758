# dev: is the fast / slow state of the dense launches the NUMA node the process (its host-side queues / kernel arguments / signals) lives on?
# prints the box's topology, then runs tools/dev/r04_state_stream.py (one engine, 2 streams) pinned to each node's cores in turn
mkdir -p gpurun_out; O=gpurun_out/state_numa.txt; : > $O
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" >> $O
for n in /sys/devices/system/node/node*; do echo "$n cpus $(cat $n/cpulist) $(grep MemFree $n/meminfo)" >> $O; done
for d in /sys/class/drm/card*/device; do echo "$d numa_node $(cat $d/numa_node 2>/dev/null) local_cpulist $(cat $d/local_cpulist 2>/dev/null) $(basename $(readlink -f $d))" >> $O; done
python - >> $O <<'PY'
import os; print('affinity of this shell:', len(os.sched_getaffinity(0)), 'cpus', sorted(os.sched_getaffinity(0))[:4], '...')
PY
NODES=$(ls -d /sys/devices/system/node/node* | sed 's/.*node//')
for r in 1 2 3; do for n in $NODES; do
  C=$(cat /sys/devices/system/node/node$n/cpulist)
  echo "== round $r node $n cpus $C" >> $O
  timeout 200 taskset -c $C python tools/dev/r04_state_stream.py 1 2 2>&1 | grep -v amdgpu.ids >> $O
done; done
cat $O
