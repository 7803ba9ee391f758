#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3s
# conv launches per step: 4 FPN, 4 RPN (side stream), 4 semantic, 4 mask: pick a level-0 semantic conv in a late step
for at in 329 330 331 333; do
NUHTC_STAMP_AT=$at timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 20 --in-flight 0 --no-settle > /dev/null 2>&1
python tools/dev/conv_stamps.py /tmp/conv_stamps.txt | tee gpurun_out/r3s/stamps_$at.txt
done
