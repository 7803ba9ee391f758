# dev: sequential vs batches in flight at several batch sizes (tiles/s, sequential | pipelined)
for cfg in "16 2" "16 3" "16 4" "8 2" "8 3" "8 4" "4 4" "32 2"; do set -- $cfg
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --batch $1 --in-flight $2 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 in-flight $2: sequential', round(d['sequential']['value'],1), '| in flight', round(d['value'],1))"
done
