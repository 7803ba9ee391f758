# dev: throughput schedule: 256-row tiles only for launches of at least NUHTC_THR_MT2_MIN such tiles (0 = always, the committed rule); bench.py in flight
export NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV
python -m nuhtc_amd.build --force > /dev/null || exit 1
mkdir -p gpurun_out; O=gpurun_out/thr_mt.txt; : > $O
for r in 1 2 3; do for v in 0 384 512 1024; do
  NUHTC_THR_MT2_MIN=$v timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 150 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('THR_MT2_MIN=$v', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'seq', round(d['sequential']['value'],1))" >> $O
done; done
cat $O
