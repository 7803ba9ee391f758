#!/bin/bash
# fused pointwise layers in the conv epilogue: parity tests, then A/B against the separate launches (dev build: NUHTC_CONV_FUSE)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fuse
if [ "$1" != "noperf" ]; then
AB_STEPS=80 bash tools/dev/r03_ab.sh NUHTC_CONV_FUSE=0 NUHTC_CONV_FUSE=1 NUHTC_CONV_FUSE=0 NUHTC_CONV_FUSE=1 NUHTC_CONV_FUSE=0 NUHTC_CONV_FUSE=1
fi
timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_full.py tests/test_hip_edges.py -m gpu -x -q > gpurun_out/fuse/tests.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/fuse/tests.log
