# dev: builds the dev library and runs r04_state_buffers.py in P processes on one box (results -> gpurun_out/state_buffers.txt)
P=${1:-2}; E=${2:-3}
export NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV
python -m nuhtc_amd.build --force > /dev/null || exit 1
mkdir -p gpurun_out; : > gpurun_out/state_buffers.txt
for i in $(seq $P); do echo "== process $i" >> gpurun_out/state_buffers.txt; timeout 400 python tools/dev/r04_state_buffers.py $E >> gpurun_out/state_buffers.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/state_buffers.txt | tail -80
