#!/bin/bash
# tile shapes judged by the rate with four batches in flight (tails are filled by other batches there)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 120"
for cfg in ${CFGS:-NUHTC_SPLIT_MT=0 NUHTC_SPLIT_MT=2 NUHTC_SPLIT_NT=6 NUHTC_SPLIT_MT=0 NUHTC_SPLIT_MT=2 NUHTC_SPLIT_NT=6}; do
  envs=$(echo "$cfg" | tr ',' ' ')
  env $envs timeout 300 $B > gpurun_out/ab/vm.json 2> gpurun_out/ab/vm.err
  python - $cfg <<'P'
import json, sys
d = json.load(open('gpurun_out/ab/vm.json'))
k = d['kernel_ms_per_step']
print(sys.argv[1], '| value %.0f seq %.0f clock %.2f' % (d['value'], d['sequential']['value'], d['roofline']['shader_clock_ghz_under_step']), {a: k.get(a) for a in ('gemm_kernel<3>', 'gemm_kernel<6>', 'gemm_kernel<2>')})
P
done
