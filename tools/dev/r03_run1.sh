#!/bin/bash
# round-3 baseline: GPU tests, bench with per-shape GEMM timings, bench at batch 64
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r3a/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/gputests.log
timeout 600 python bench.py --gemm-shapes > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
timeout 400 python bench.py --batch 64 --steps 30 --no-cpu-baseline --no-fp32-pipe --no-roi-load > gpurun_out/r3a/bench_b64.json 2> gpurun_out/r3a/bench_b64.err
tail -5 gpurun_out/r3a/gputests.log
