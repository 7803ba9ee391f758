# dev: per-shape GEMM table of the bench step (default pipe), one line per shape sorted by time
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 50 --gemm-shapes 2>/dev/null > /tmp/shapes.json
python - <<'PY'
import json
d = json.load(open('/tmp/shapes.json'))
print(round(d['value'], 1), d['ms_per_step'])
g = d['gemm_shapes']
tot = 0
for k, v in sorted(g.items(), key=lambda kv: -kv[1]['ms_per_step']):
    tot += v['ms_per_step']
    print(f"{k:58s} ms={v['ms_per_step']:.3f} tf={v.get('tflops', 0):7.1f}")
print('total', round(tot, 3))
PY
