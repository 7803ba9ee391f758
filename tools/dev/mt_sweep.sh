# dev: split GEMM block-tile forms over the bench step
for f in 1 2 0; do
  NUHTC_SPLIT_MT=$f python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 50 --gemm-shapes 2>/dev/null > /tmp/mt_$f.json
  python - $f <<'PY'
import json, sys
f = sys.argv[1]
d = json.load(open(f'/tmp/mt_{f}.json'))
g = d['gemm_shapes']
print('MT', f, round(d['value'], 1), round(d['ms_per_step'], 3), {k.split('<')[1]: v['tflops'] for k, v in sorted(g.items(), key=lambda kv: -kv[1]['ms_per_step'])[:16]})
PY
done
