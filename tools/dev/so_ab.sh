# dev: A/B two builds (tmp_ab/<a>.so, tmp_ab/<b>.so) on the bench step, alternating; prints sequential rate and the named kernel tags
A=$1; B=$2; TAGS=${3:-gemm_kernel<3>,gemm_kernel<2>,gemm_kernel<1>}
cp nuhtc_amd/libnuhtc_hip.so /tmp/keep.so
for r in 1 2 3 4; do for v in $A $B; do cp tmp_ab/$v.so nuhtc_amd/libnuhtc_hip.so; python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$v', 'sequential', round(d['value'],1), round(d['roofline']['frac'],4), {t: k.get(t) for t in '$TAGS'.split(',')})"; done; done
cp /tmp/keep.so nuhtc_amd/libnuhtc_hip.so
