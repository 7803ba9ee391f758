# dev: sweep of the GEMM column-tile heuristic (NUHTC_GEMM_FILL) over the bench step
for f in 500 300 200 800; do
  NUHTC_GEMM_FILL=$f python bench.py --no-cpu-baseline --no-roi-load --steps 50 --gemm-shapes 2>/dev/null > /tmp/fill_$f.json
  python - $f <<'PY'
import json, sys
f = sys.argv[1]
d = json.load(open(f'/tmp/fill_{f}.json'))
g = d['gemm_shapes']
print(f, round(d['value'], 1), round(d['ms_per_step'], 3), {k: v['ms_per_step'] for k, v in g.items() if 'K3136' in k or 'N768|K3072' in k or 'N64|K576' in k or 'N256|K256' in k})
PY
done
