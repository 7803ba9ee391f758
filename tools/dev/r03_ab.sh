#!/bin/bash
# quick A/B: env settings given as arguments "NAME=VAL,NAME2=VAL2" ... one bench per argument (dev build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps ${AB_STEPS:-40} --gemm-shapes"
i=0
for cfg in "$@"; do
  i=$((i+1))
  envs=$(echo "$cfg" | tr ',' ' ')
  env $envs timeout 300 $B > gpurun_out/ab/$i.json 2> gpurun_out/ab/$i.err
  python - "$cfg" gpurun_out/ab/$i.json <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d['kernel_ms_per_step']
    print(sys.argv[1], '| value %.0f seq %.0f (%.2f ms) clock %.2f' % (d['value'], d['sequential']['value'], d['sequential']['ms_per_step'], d['roofline']['shader_clock_ghz_under_step']),
          {a: k.get(a) for a in ('gemm_kernel<3>', 'gemm_kernel<2>', 'gemm_kernel<1>', 'swin_mlp', 'window_attn', 'layernorm')})
    print('   ', {a: b['ms_per_step'] for a, b in d['gemm_shapes'].items() if 'conv3' in a or 'K96' in a or 'N256' in a or 'N1024' in a})
except Exception as e:
    print(sys.argv[1], 'failed', e)
P
done
