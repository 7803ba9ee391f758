# dev: two late round-5 changes of the Swin stages, each against the form it replaces, one process alternating the settings:
#   ATTN_PADBITS  1 (the tree): the attention kernel never reads a padding row of the window image (one bias row stands in) / 0: the rows written and read
#   MERGE_LN_IN_A 1 (the tree): the PatchMerging norm in the A path of its reduction linear (two-segment rows) / 0: merge_ln_kernel + plain GEMM
# Parity tests first, then the sequential step and tags per setting, the bench's in-flight rate per setting, and the Swin traffic of the tree.
mkdir -p gpurun_out; O=gpurun_out/r05_padbits.txt; : > $O
timeout 1200 python -m pytest tests/test_hip_dense.py tests/test_hip_full.py tests/test_hip_edges.py -m gpu -x -q 2>&1 | tail -3 >> $O
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
timeout 300 python tools/dev/knob_ab.py ATTN_PADBITS 0 1 --rounds 12 --tags gemm,window_attn,layernorm,swin_lnqkv >> $O 2>/dev/null
timeout 300 python tools/dev/knob_ab.py MERGE_LN_IN_A 0 1 --rounds 12 --tags gemm,merge_ln,layernorm,swin_mlp >> $O 2>/dev/null
for r in 1 2; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  NUHTC_ATTN_PADBITS=$1 NUHTC_MERGE_LN_IN_A=$2 timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('ATTN_PADBITS=$1 MERGE_LN_IN_A=$2 in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'window_attn', k.get('window_attn'), 'gemm<3>', k.get('gemm_kernel<3>'), 'merge_ln', k.get('merge_ln'), 'clock', d['roofline']['shader_clock_ghz_under_step'])" >> $O
done; done
python -m nuhtc_amd.build --force > /dev/null
cat $O
bash tools/dev/r05_traffic_only.sh 2>&1 | tail -18
