#!/bin/bash
# batches in flight at 16 hardware queues, engines on their own streams
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/q
for d in ${DEPTHS:-4 5 6 3 4 6}; do
  timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 120 --in-flight $d > gpurun_out/q/d$d.json 2> gpurun_out/q/d$d.err
  python - $d <<'P'
import json,sys
d=json.load(open('gpurun_out/q/d%s.json'%sys.argv[1])); print('depth', sys.argv[1], 'value %.0f seq %.0f clock %.2f' % (d['value'], d['sequential']['value'], d['roofline']['shader_clock_ghz_under_step']))
P
done
