#!/bin/bash
# fused MLP: correctness, then A/B in one process-per-setting (dev build: knobs from the environment)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_hip_dense.py -m gpu -q -x -s > gpurun_out/r3b/dense.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3b/dense.log
tail -3 gpurun_out/r3b/dense.log
timeout 600 python -m pytest tests/test_hip_full.py -m gpu -q -x > gpurun_out/r3b/full.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3b/full.log
tail -3 gpurun_out/r3b/full.log
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 40 --gemm-shapes"
NUHTC_FUSED_MLP=0 timeout 300 $B > gpurun_out/r3b/b_unfused.json 2> gpurun_out/r3b/b_unfused.err
NUHTC_FUSED_MLP=1 NUHTC_MLP_STAGGER=0 timeout 300 $B > gpurun_out/r3b/b_fused_nostag.json 2> gpurun_out/r3b/b_fused_nostag.err
NUHTC_FUSED_MLP=1 NUHTC_MLP_STAGGER=1 timeout 300 $B > gpurun_out/r3b/b_fused_stag.json 2> gpurun_out/r3b/b_fused_stag.err
NUHTC_FUSED_MLP=0 timeout 300 $B > gpurun_out/r3b/b_unfused2.json 2> gpurun_out/r3b/b_unfused2.err
python - <<'P'
import json
for n in ('b_unfused','b_fused_nostag','b_fused_stag','b_unfused2'):
    try:
        d=json.load(open(f'gpurun_out/r3b/{n}.json'))
        k=d['kernel_ms_per_step']
        print(n, 'value %.0f seq %.0f (%.2f ms) clock %.2f'%(d['value'], d['sequential']['value'], d['sequential']['ms_per_step'], d['roofline']['shader_clock_ghz_under_step']), 'gemm3', k.get('gemm_kernel<3>'), 'mlp', k.get('swin_mlp'), 'ln', k.get('layernorm'))
    except Exception as e: print(n, 'failed', e)
P
