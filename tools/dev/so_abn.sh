# dev: A/B/... of N builds (tmp_ab/<name>.so ...) on the bench step, alternating processes on ONE box; prints the sequential rate, the
# dominant kernel's fraction and the named kernel tags per run.   usage: bash tools/dev/so_abn.sh "old v1 v2" [rounds] [tags]
VARS=$1; ROUNDS=${2:-3}; TAGS=${3:-gemm_kernel<3>,gemm_kernel<2>,gemm_kernel<1>}
cp nuhtc_amd/libnuhtc_hip.so /tmp/keep.so
for r in $(seq $ROUNDS); do for v in $VARS; do cp tmp_ab/$v.so nuhtc_amd/libnuhtc_hip.so; python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$v', 'sequential', round(d['value'],1), round(d['roofline']['frac'],4), d['roofline']['shader_clock_ghz_under_step'], {t: k.get(t) for t in '$TAGS'.split(',')})"; done; done
cp /tmp/keep.so nuhtc_amd/libnuhtc_hip.so
