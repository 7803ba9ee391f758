# dev: A/B an environment switch on the bench step, alternating processes: env_ab.sh VAR v0 v1 [tags]
V=$1; A=$2; B=$3; TAGS=${4:-cc_proposals,nms,rpn_select}
for r in 1 2 3; do for v in $A $B; do env $V=$v python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('$V=$v', round(d['value'],1), round(d['ms_per_step'],3), {t: k.get(t) for t in '$TAGS'.split(',')})"; done; done
