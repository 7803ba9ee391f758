# dev: batches in flight (tiles/s in flight | sequential), B = 16
for d in 3 4 5 6 8 4; do
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight $d --steps 120 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $d:', round(d['value'],1), '| sequential', round(d['sequential']['value'],1))"
done
