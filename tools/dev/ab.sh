# dev helper: A/B two builds of libnuhtc_hip.so (tmp_ab/old.so, tmp_ab/new.so) on one box, interleaved
for v in old new old new old new; do
  cp tmp_ab/$v.so nuhtc_amd/libnuhtc_hip.so
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$v', round(d['value'],1), 'tiles/s', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4), {n: k.get(n) for n in ('gemm_kernel<3>','gemm_kernel<2>','conv1x1_n1','paste','roi_feat14')})"
done
