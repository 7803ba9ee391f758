# dev: column-tile narrowing of the split GEMMs (NUHTC_SPLIT_FILL: narrow until the launch has that many workgroups; 256 = committed, 0 = never)
# with four batches in flight -- does a launch that covers part of the chip still want the wider tile there, as the 256-row sweep says?
export NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV
python -m nuhtc_amd.build --force > /dev/null || exit 1
mkdir -p gpurun_out; O=gpurun_out/fill_ab.txt; : > $O
for r in 1 2 3; do for v in 256 128 0; do
  NUHTC_SPLIT_FILL=$v timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 150 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('SPLIT_FILL=$v', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'seq', round(d['sequential']['value'],1), {t: k.get(t) for t in ('gemm_kernel<1>','gemm_kernel<2>','gemm_kernel<4>')})" >> $O
done; done
cat $O
