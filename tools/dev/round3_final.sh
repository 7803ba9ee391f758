#!/bin/bash
# end-of-round artefacts on the default build: strict parity log, the measured summaries of round3_all.sh, the slide-level bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/round3
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/round3/r03_parity_gpu_strict.log 2>&1; echo "pytest rc $?" >> gpurun_out/round3/r03_parity_gpu_strict.log
tail -3 gpurun_out/round3/r03_parity_gpu_strict.log
bash tools/dev/round3_all.sh
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/bench_wsi.py > gpurun_out/round3/r03_bench_wsi.json 2> gpurun_out/round3/r03_bench_wsi.err; tail -c 600 gpurun_out/round3/r03_bench_wsi.json
