#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c2
timeout 1500 python -m pytest tests/test_hip_full.py tests/test_hip_edges.py tests/test_wsi_canvas.py -m gpu -q -x > gpurun_out/r3c2/tests.log 2>&1; tail -3 gpurun_out/r3c2/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 40 > gpurun_out/r3c2/b.json 2> gpurun_out/r3c2/b.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r3c2/b.json'))
print('value %.0f seq %.0f (%.2f ms) clock %.2f'%(d['value'], d['sequential']['value'], d['sequential']['ms_per_step'], d['roofline']['shader_clock_ghz_under_step']))
print(d['kernel_ms_per_step'])
P
