#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab gpurun_out/q
for q in 4 16 8 4 16 8; do echo "queues $q"; GPU_MAX_HW_QUEUES=$q DEPTHS="4" bash tools/dev/r03_depth.sh; done
echo "queues 4"; GPU_MAX_HW_QUEUES=4 DEPTHS="3 5" bash tools/dev/r03_depth.sh
