# dev: LayerNorm of Swin stages 2-4.  LN_IN_A=0: the norms as kernels of their own (round 4); 1: in the A path of the QKV / fc1 linears, statistics by
# ln_stats_kernel; 2 (the tree): statistics left by the epilogue of the GEMM that produced the tensor.  One process, alternating settings (sequential
# step + the GEMM and layernorm tags), then the bench's in-flight rate per setting.
mkdir -p gpurun_out; O=gpurun_out/r05_ln_in_a.txt; : > $O
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
timeout 300 python tools/dev/knob_ab.py LN_IN_A 0 1 2 --rounds 9 --tags gemm,layernorm >> $O 2>/dev/null
for r in 1 2; do for v in 0 1 2; do
  NUHTC_LN_IN_A=$v timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('LN_IN_A=$v in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'gemm<3>', k.get('gemm_kernel<3>'), 'layernorm', k.get('layernorm'), 'clock', d['roofline']['shader_clock_ghz_under_step'])" >> $O
done; done
python -m nuhtc_amd.build --force > /dev/null
cat $O
