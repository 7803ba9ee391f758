# dev: under the throughput schedule every 96-column launch takes 256-row tiles; the stage-4 proj / fc2 and the last merging linear are then 128
# workgroups on 256 CUs.  SPLIT_MT2_MIN = least number of 256-row tiles for which that form is taken (0 = the tree).  The knob is NOT in the tree (no gain:
# profiles/r05_mt2_min.txt); the patch it needs in launch_gemm (gemm.hip), in front of `const bool mt2 = ...`:
#     static const int& mt2_min = dev_knob_ref("SPLIT_MT2_MIN", 0);
#     ... (p.throughput && blocks2 >= mt2_min) instead of p.throughput in the mt2 expression
mkdir -p gpurun_out; O=gpurun_out/r05_mt2_min.txt; : > $O
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
for r in 1 2 3; do for v in 0 192 300 600; do
  NUHTC_SPLIT_MT2_MIN=$v timeout 300 python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SPLIT_MT2_MIN=$v in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'clock', round(d['roofline']['shader_clock_ghz_under_step'],3))" >> $O
done; done
python -m nuhtc_amd.build --force > /dev/null
cat $O
