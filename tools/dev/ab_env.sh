# dev helper: A/B an environment switch on one box, interleaved: ab_env.sh VAR valueA valueB [valueC]
for i in 1 2 3; do for v in $2 $3 $4; do
  env $1=$v python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$1=$v', round(d['value'],1), 'tiles/s', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), {n: k[n] for n in ('gemm_kernel<3>','gemm_kernel<2>','gemm_kernel<1>')})"
done; done
