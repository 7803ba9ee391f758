# dev (round 5): room for other batches' memory-bound kernels beside the split GEMM.  The GEMM's workgroups fill every CU's registers (2 x 4 waves
# x 235, or 3 x 4 x 157), so kernels of different batches alternate on a CU instead of sharing it (round 4's ablation: the step in flight is the SUM
# of all kernel groups).  NUHTC_GEMM_LDS_PAD gives the GEMM dynamic LDS it never touches: with 128-row tiles (NUHTC_SPLIT_MT=1, 46 KB of LDS, 157
# registers) a pad of 16 KB lets only TWO of its workgroups onto a CU and leaves a third of the registers and ~36 KB of LDS to whoever else has work.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_reserve.txt; : > $O
NUHTC_EXTRA_CFLAGS="-DNUHTC_DEV" python -m nuhtc_amd.build --force > /dev/null 2>&1 || exit 1
run() { python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'clock', round(d['roofline']['shader_clock_ghz_under_step'] or 0, 3))" >> $O; }
for r in 1 2; do
  run "tree (256-row tiles in flight)"
  NUHTC_SPLIT_MT=1 run "128-row tiles, 3 per CU"
  NUHTC_SPLIT_MT=1 NUHTC_GEMM_LDS_PAD=16384 run "128-row tiles, pad 16 KB -> 2 per CU"
  NUHTC_GEMM_LDS_PAD=40960 run "256-row tiles, pad 40 KB -> 1 per CU"
done
python -m nuhtc_amd.build --force > /dev/null 2>&1
cat $O
