#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 40 --gemm-shapes"
for cfg in NUHTC_SPLIT_NT=0 NUHTC_SPLIT_NT=6 NUHTC_SPLIT_NT=0 NUHTC_SPLIT_NT=6; do
  env $cfg timeout 300 $B > gpurun_out/ab/nt.json 2> gpurun_out/ab/nt.err
  python - $cfg <<'P'
import json, sys
d = json.load(open('gpurun_out/ab/nt.json'))
k = d['kernel_ms_per_step']
print(sys.argv[1], '| value %.0f seq %.0f clock %.2f' % (d['value'], d['sequential']['value'], d['roofline']['shader_clock_ghz_under_step']), {a: k.get(a) for a in ('gemm_kernel<3>', 'gemm_kernel<6>', 'gemm_kernel<2>')})
print('   ', {a.replace('gemm_kernel', 'g'): b['ms_per_step'] for a, b in sorted(d['gemm_shapes'].items()) if ('<3>' in a or '<6>' in a)})
P
done
