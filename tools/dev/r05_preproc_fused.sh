# dev: resize + Normalize + Pad inside the patch embedding (PREPROC_FUSED=1, the tree) against preproc_kernel + patch_embed_kernel (0).
mkdir -p gpurun_out; O=gpurun_out/r05_preproc_fused.txt; : > $O
timeout 1200 python -m pytest tests/test_hip_dense.py tests/test_hip_full.py tests/test_hip_edges.py -m gpu -x -q 2>&1 | tail -3 >> $O
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
timeout 300 python tools/dev/knob_ab.py PREPROC_FUSED 0 1 --rounds 12 --tags preproc,patch_embed >> $O 2>/dev/null
for r in 1 2 3; do for v in 0 1; do
  NUHTC_PREPROC_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('PREPROC_FUSED=$v in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'preproc', k.get('preproc'), 'patch_embed', k.get('patch_embed'), 'clock', round(d['roofline']['shader_clock_ghz_under_step'],3))" >> $O
done; done
python -m nuhtc_amd.build --force > /dev/null
cat $O
