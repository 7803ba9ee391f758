#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for sz in 40,100 60,120; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 20 --in-flight 0 --fixed-load --roi-size $sz > /tmp/b.json 2> /tmp/b.err
  python - $sz <<'P'
import json, sys
d = json.load(open('/tmp/b.json'))
print('sizes', sys.argv[1], 'step %.2f ms  roi_feat7 %.3f ms' % (d['ms_per_step'], d['kernel_ms_per_step'].get('roi_feat7', 0)))
P
done
timeout 600 python -m pytest tests/test_hip_full.py -m gpu -q -x -k "roi" 2>&1 | tail -2
