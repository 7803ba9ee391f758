# dev: round6_all.sh + the whole GPU suite with its printed parity details (-> gpurun_out/round6/r06_parity_gpu_strict.log)
bash tools/dev/round6_all.sh > gpurun_out/round6_all.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/round6/r06_parity_gpu_strict.log
tail -3 gpurun_out/round6/r06_parity_gpu_strict.log
tail -c 600 gpurun_out/round6/r06_bench.json
