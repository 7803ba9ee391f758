#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_hip_api.py -m gpu -x -q -k "tile_policy or engine_stream or launches_its_own" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/ab/tp.json 2> gpurun_out/ab/tp.err
python - <<'P'
import json
d=json.load(open('gpurun_out/ab/tp.json'))
print('value %.0f seq %.0f roi_load %.0f fp32 %.0f clock %.2f' % (d['value'], d['sequential']['value'], d['real_slide_roi_load']['value'], d['fp32_mfma_pipe']['value'], d['roofline']['shader_clock_ghz_under_step']))
P
done
