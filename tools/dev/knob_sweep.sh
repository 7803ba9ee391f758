# dev: A/B of an environment knob on the bench step (interleaved): knob_sweep.sh "A=0" "NUHTC_SPLIT_WIDE=256" ...
run() { env $1 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {n: k.get(n) for n in ('gemm_kernel<3>','gemm_kernel<6>','gemm_kernel<2>','gemm_kernel<1>','gemm_kernel<4>')})"; }
for r in 1 2 3; do for c in "$@"; do run "$c"; done; done
