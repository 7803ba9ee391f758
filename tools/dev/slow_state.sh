#!/bin/bash
# Dev: consecutive processes on one box -- duration of the Swin linears, probe clock, and what rocm-smi says between them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ss
smi() { rocm-smi --showtemp --showpower --showclocks --showperflevel 2>/dev/null | grep -E "Temperature|Power|sclk|mclk|fclk|socclk|Performance" | tr -s ' ' | tr '\n' ';' | cut -c1-700; echo; }
smi
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 60 --in-flight 0 > gpurun_out/ss/b.json 2>/dev/null
  python - <<'P'
import json
d=json.load(open('gpurun_out/ss/b.json')); k=d['kernel_ms_per_step']
print('seq %.0f clock %.2f gemm<3> %.3f gemm<2> %.3f attn %.3f LN %.3f' % (d['value'], d['roofline']['shader_clock_ghz_under_step'], k['gemm_kernel<3>'], k['gemm_kernel<2>'], k['window_attn'], k['layernorm']))
P
  smi
  if [ $i = 3 ]; then sleep 45; echo "(slept 45 s)"; smi; fi
done
