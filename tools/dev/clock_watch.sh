# dev: sample rocm-smi clocks / power while the bench step runs
(python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 1500 > /tmp/cw_bench.json 2>/dev/null) &
BP=$!
while kill -0 $BP 2>/dev/null; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|Package Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.3; done | sort | uniq -c | sort -k1 -n -r | head -25
python -c "import json; d=json.load(open('/tmp/cw_bench.json')); print(d['value'], d['ms_per_step'])"
