#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_full.py -m gpu -q -x -k "roi or size" 2>&1 | tail -3
for sz in 60,120 100,200 150,400; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 20 --in-flight 0 --fixed-load --roi-size $sz > /tmp/b.json 2> /tmp/b.err
  python - $sz <<'P'
import json, sys
d = json.load(open('/tmp/b.json'))
print('sizes', sys.argv[1], 'step %.2f ms  roi_feat7 %.3f ms' % (d['ms_per_step'], d['kernel_ms_per_step'].get('roi_feat7', 0)))
P
done
timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --steps 30 > /tmp/b.json 2> /tmp/b.err
python - <<'P'
import json
d=json.load(open('/tmp/b.json'))
print('value %.0f seq %.0f (%.2f ms) roi_load %.0f clock %.2f roi_feat7 %.3f'%(d['value'], d['sequential']['value'], d['sequential']['ms_per_step'], d['real_slide_roi_load']['value'], d['roofline']['shader_clock_ghz_under_step'], d['kernel_ms_per_step']['roi_feat7']))
P
