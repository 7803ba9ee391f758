# dev: the split GEMM with persistent workgroups (NUHTC_SPLIT_PERSIST=1, the default of this build) against one workgroup per tile (=0):
# one process alternating (knob_ab.py, sequential step + GEMM tags), then bench.py processes alternating (four batches in flight)
export NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV
python -m nuhtc_amd.build --force > /dev/null || exit 1
mkdir -p gpurun_out; O=gpurun_out/persist_ab.txt; : > $O
timeout 300 python tools/dev/knob_ab.py SPLIT_PERSIST 0 1 --rounds 10 2>&1 | grep -v amdgpu.ids >> $O
for r in 1 2 3; do for v in 0 1; do
  NUHTC_SPLIT_PERSIST=$v timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('persist=$v', 'value', round(d['value'],1), 'seq', round(d['sequential']['value'],1), 'gemm3', k['gemm_kernel<3>'], 'gemm2', k['gemm_kernel<2>'], 'frac', round(d['roofline']['frac'],4))" >> $O
done; done
cat $O
