# dev: builds the dev library and runs r04_ablate.py for two RoI loads (results -> gpurun_out/ablate.txt)
export NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV
python -m nuhtc_amd.build --force > /dev/null || exit 1
mkdir -p gpurun_out; : > gpurun_out/ablate.txt
for sz in 12,40 40,100; do timeout 400 python tools/dev/r04_ablate.py $sz 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.txt; done
cat gpurun_out/ablate.txt
