#!/bin/bash
# full GPU suite + default bench on the default (non-dev) build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3full
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3full/gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3full/gputests.log
tail -6 gpurun_out/r3full/gputests.log
timeout 600 python bench.py --gemm-shapes > gpurun_out/r3full/bench.json 2> gpurun_out/r3full/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r3full/bench.json'))
print('value %.0f seq %.0f roi_load %.0f frac %.3f pipe %.3f clock %.2f'%(d['value'], d['sequential']['value'], d['real_slide_roi_load']['value'], d['roofline']['frac'], d['roofline']['matrix_pipe']['frac'], d['roofline']['shader_clock_ghz_under_step']))
print(d['kernel_groups'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['parity']['passed'])
P
