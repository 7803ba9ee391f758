# dev: round5_all.sh + the whole GPU suite with its printed parity details (-> gpurun_out/round5/r05_parity_gpu_strict.log)
bash tools/dev/round5_all.sh > gpurun_out/round5_all.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/round5/r05_parity_gpu_strict.log
tail -3 gpurun_out/round5/r05_parity_gpu_strict.log
