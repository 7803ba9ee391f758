#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/q
for q in 4 8 16; do NENG=8 GPU_MAX_HW_QUEUES=$q timeout 250 python tools/dev/queue_map.py 2>&1 | tail -2; done
for q in 4 16 4 16 8 24; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --steps 80 > gpurun_out/q/b$q.json 2> gpurun_out/q/b$q.err
  python - $q <<'P'
import json,sys
d=json.load(open('gpurun_out/q/b%s.json'%sys.argv[1]))
print('queues', sys.argv[1], d['config'].get('hw_queues'), 'value %.0f seq %.0f roi_load %.0f fp32 %.0f clock %.2f' % (d['value'], d['sequential']['value'], d['real_slide_roi_load']['value'], d['fp32_mfma_pipe']['value'], d['roofline']['shader_clock_ghz_under_step']))
P
done
