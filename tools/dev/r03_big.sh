#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/big
timeout 900 python -m pytest tests/test_hip_full.py -m gpu -x -q -k "roi or giant or full_path" > gpurun_out/big/tests.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/big/tests.log
AB_STEPS=60 bash tools/dev/r03_ab.sh NUHTC_ROI_LDS_ORDER=0 NUHTC_ROI_LDS_ORDER=1 NUHTC_ROI_LDS_ORDER=0 NUHTC_ROI_LDS_ORDER=1
for i in 1 2 3 4; do python - gpurun_out/ab/$i.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print({k:v for k,v in d['kernel_ms_per_step'].items() if 'roi' in k})
P
done
for rs in 40,100 100,200; do
for fz in 0 1; do
NUHTC_STREAM_FEW=$fz NUHTC_BIG_SPLIT=$fz timeout 300 python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 30 --in-flight 0 --fixed-load --roi-size $rs > gpurun_out/big/fl.json 2> gpurun_out/big/fl.err
python - $rs $fz <<'P'
import json,sys
d=json.load(open('gpurun_out/big/fl.json')); print('roi-size', sys.argv[1], 'few/split', sys.argv[2], 'value %.0f' % d['value'], {k:v for k,v in d['kernel_ms_per_step'].items() if 'roi' in k})
P
done; done
