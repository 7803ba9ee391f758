#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/nt
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 30 --gemm-shapes --in-flight 0"
for cfg in NUHTC_SPLIT_NT=0 NUHTC_SPLIT_NT=4 NUHTC_SPLIT_NT=2 NUHTC_SPLIT_NT=1 NUHTC_SPLIT_NT=0; do
  env $cfg timeout 300 $B > gpurun_out/nt/$cfg.json 2> /dev/null
done
python - <<'P'
import json
d = {c: json.load(open(f'gpurun_out/nt/NUHTC_SPLIT_NT={c}.json')) for c in (0, 4, 2, 1)}
print({c: round(d[c]['ms_per_step'], 2) for c in d})
keys = set()
for c in d:
    keys |= {k.split('|', 1)[1] for k in d[c]['gemm_shapes']}
def get(c, sh):
    for k, v in d[c]['gemm_shapes'].items():
        if k.split('|', 1)[1] == sh:
            return f"{k.split('|')[0][-2]}:{v['ms_per_step']:.3f}"
    return '-'
for sh in sorted(keys):
    print(f"{sh:22s} " + '  '.join(f"nt{c}={get(c, sh)}" for c in d))
P
