#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_full.py -m gpu -q -x -s -k "roi or size" > gpurun_out/r3r/tests2.log 2>&1; grep -E "RoIs with|boxes |mask RoIs|passed|failed|Error" gpurun_out/r3r/tests2.log | tail -12
