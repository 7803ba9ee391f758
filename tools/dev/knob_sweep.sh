# dev: A/B of the split-kernel tile-selection knobs on the bench step (interleaved, two rounds)
run() { env "$@" python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$*', round(d['value'],1), round(d['ms_per_step'],3), {n: k.get(n) for n in ('gemm_kernel<3>','gemm_kernel<2>','gemm_kernel<1>','gemm_kernel<4>')})"; }
for r in 1 2; do
run A=0
run NUHTC_SPLIT_KEEP3=1
run NUHTC_SPLIT_KEEP3=1 NUHTC_SPLIT_FILL=256
run NUHTC_SPLIT_KEEP3=1 NUHTC_SPLIT_FILL=128
run NUHTC_SPLIT_KEEP3=1 NUHTC_SPLIT_FILL=256 NUHTC_SPLIT_MT2NT2=1
run NUHTC_SPLIT_MT2NT2=1
done
