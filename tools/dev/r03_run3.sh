#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 20 --in-flight 0"
NUHTC_MLP_STAGGER=0 NUHTC_STAMP_AT=60 timeout 300 $B > gpurun_out/r3c/b0.json 2> gpurun_out/r3c/b0.err; cp /tmp/mlp_stamps.txt gpurun_out/r3c/stamps_nostag.txt
NUHTC_MLP_STAGGER=1 NUHTC_STAMP_AT=60 timeout 300 $B > gpurun_out/r3c/b1.json 2> gpurun_out/r3c/b1.err; cp /tmp/mlp_stamps.txt gpurun_out/r3c/stamps_stag.txt
python tools/dev/mlp_stamps.py gpurun_out/r3c/stamps_nostag.txt; python tools/dev/mlp_stamps.py gpurun_out/r3c/stamps_stag.txt
