#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for d in 4 3 5 6 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 60 --in-flight $d > /tmp/b.json 2>/dev/null
  python - $d <<'P'
import json, sys
d = json.load(open('/tmp/b.json'))
print('in-flight', sys.argv[1], 'value %.0f seq %.0f clock %.2f' % (d['value'], d['sequential']['value'], d['roofline']['shader_clock_ghz_under_step']))
P
done
