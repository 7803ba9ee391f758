#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3e
NUHTC_CONV_HALO=0 timeout 200 python tools/dev/dump_stage.py /tmp/a.npz > gpurun_out/r3e/dump.log 2>&1
NUHTC_CONV_HALO=1 timeout 200 python tools/dev/dump_stage.py /tmp/b.npz >> gpurun_out/r3e/dump.log 2>&1
python tools/dev/cmp_stage.py /tmp/a.npz /tmp/b.npz | grep -c "identical True"
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 40 --gemm-shapes"
NUHTC_CONV_HALO=0 timeout 300 $B > gpurun_out/r3e/b_old.json 2> gpurun_out/r3e/b_old.err
NUHTC_CONV_HALO=1 timeout 300 $B > gpurun_out/r3e/b_halo.json 2> gpurun_out/r3e/b_halo.err
python - <<'P'
import json
for n in ('b_old','b_halo'):
    try:
        d=json.load(open(f'gpurun_out/r3e/{n}.json'))
        k=d['kernel_ms_per_step']
        print(n, 'value %.0f seq %.0f (%.2f ms) clock %.2f'%(d['value'], d['sequential']['value'], d['sequential']['ms_per_step'], d['roofline']['shader_clock_ghz_under_step']), 'gemm2', k.get('gemm_kernel<2>'), 'gemm1', k.get('gemm_kernel<1>'))
        print({a:b for a,b in d['gemm_shapes'].items() if 'conv3' in a})
    except Exception as e: print(n, 'failed', e)
P
bash tools/dev/r03_prof.sh
