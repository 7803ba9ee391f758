#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_full.py -m gpu -q -x > gpurun_out/r3h/tests.log 2>&1; tail -3 gpurun_out/r3h/tests.log
bash tools/dev/r03_ab.sh NUHTC_X=0 NUHTC_X=1
