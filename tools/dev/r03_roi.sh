#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_full.py -m gpu -q -x -k "roi or full_path or size" > gpurun_out/r3r/tests.log 2>&1; tail -4 gpurun_out/r3r/tests.log
for sz in 12,40 40,100 60,120 100,200 150,400; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 20 --in-flight 0 --fixed-load --roi-size $sz > gpurun_out/r3r/b_$sz.json 2> gpurun_out/r3r/b_$sz.err
  python - $sz <<'P'
import json, sys
try:
    d = json.load(open(f'gpurun_out/r3r/b_{sys.argv[1]}.json'))
    print(sys.argv[1], 'step %.2f ms  roi_feat7 %.3f ms' % (d['ms_per_step'], d['kernel_ms_per_step'].get('roi_feat7', 0)))
except Exception as e:
    print(sys.argv[1], 'failed', e)
P
done
