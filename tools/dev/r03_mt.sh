#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/mt
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 30 --gemm-shapes --in-flight 0"
for cfg in NUHTC_SPLIT_MT=0 NUHTC_SPLIT_MT=1 NUHTC_SPLIT_MT=2 NUHTC_SPLIT_MT=0; do
  env $cfg timeout 300 $B > gpurun_out/mt/$cfg.json 2> /dev/null
done
python - <<'P'
import json
d = {c: json.load(open(f'gpurun_out/mt/NUHTC_SPLIT_MT={c}.json')) for c in (0, 1, 2)}
print({c: round(d[c]['ms_per_step'], 2) for c in d})
for k in sorted(d[0]['gemm_shapes'], key=lambda k: -d[0]['gemm_shapes'][k]['ms_per_step']):
    if k.startswith('gemm_kernel<3>'):
        print(f"{k:34s} auto {d[0]['gemm_shapes'][k]['ms_per_step']:.3f}  mt1 {d[1]['gemm_shapes'].get(k, {}).get('ms_per_step', 0):.3f}  mt2 {d[2]['gemm_shapes'].get(k, {}).get('ms_per_step', 0):.3f}")
P
