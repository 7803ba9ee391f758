#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab gpurun_out/q
timeout 600 python -m pytest tests/test_hip_api.py -m gpu -x -q -k "schedule or engine_stream or launches_its_own or two_ranks or batch_64" 2>&1 | tail -3
DEPTHS="4 5 6 8 4 5 6" bash tools/dev/r03_depth.sh
GPU_MAX_HW_QUEUES=4 DEPTHS="4 6" bash tools/dev/r03_depth.sh
