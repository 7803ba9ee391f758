# dev: node x HIP_FORCE_DEV_KERNARG matrix of r04_state_numa2.py, then processes that move between the nodes after start
mkdir -p gpurun_out; O=gpurun_out/state_numa2.txt; : > $O
C0=$(cat /sys/devices/system/node/node0/cpulist); C1=$(cat /sys/devices/system/node/node1/cpulist)
for r in 1 2; do for k in unset 0 1; do for c in "$C0" "$C1"; do
  if [ $k = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$k; fi
  timeout 200 taskset -c $c python tools/dev/r04_state_numa2.py 2>&1 | grep -v amdgpu.ids >> $O
done; done; done
unset HIP_FORCE_DEV_KERNARG
echo "== moves" >> $O
MOVE=$C0 timeout 200 taskset -c $C1 python tools/dev/r04_state_numa2.py 2>&1 | grep -v amdgpu.ids >> $O
MOVE=$C1 timeout 200 taskset -c $C0 python tools/dev/r04_state_numa2.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
