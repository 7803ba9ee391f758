#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3g
NUHTC_FUSED_QKV=0 timeout 200 python tools/dev/dump_stage.py /tmp/a.npz > gpurun_out/r3g/dump.log 2>&1
NUHTC_FUSED_QKV=1 timeout 200 python tools/dev/dump_stage.py /tmp/b.npz >> gpurun_out/r3g/dump.log 2>&1
tail -2 gpurun_out/r3g/dump.log
python tools/dev/cmp_stage.py /tmp/a.npz /tmp/b.npz | tee gpurun_out/r3g/cmp.txt
timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_full.py -m gpu -q -x > gpurun_out/r3g/tests.log 2>&1; tail -3 gpurun_out/r3g/tests.log
bash tools/dev/r03_ab.sh NUHTC_FUSED_QKV=0 NUHTC_FUSED_QKV=1
