# One GPU call that refreshes the measured artefacts of round 3 (run through gpurun from the repo root, default build):
#   bash tools/dev/round3_all.sh
R=r03
OUT=$GRAFT_REPO_ROOT/gpurun_out/round3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/ut /tmp/hf /tmp/hw
P="--steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle --no-fp32-pipe --in-flight 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/hf -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/hw -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' 'gemm_split_kernel<[12], 3, 0>' $GRAFT_REPO_ROOT/nuhtc_amd/csrc/gemm.hip > $OUT/${R}_traffic.json; cat $OUT/${R}_traffic.json
cp $OUT/${R}_traffic.json $GRAFT_REPO_ROOT/profiles/${R}_traffic.json
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_hbm_all.py '/tmp/h[fw]/**/*counter_collection.csv' 30 > $OUT/${R}_hbm_per_kernel.txt
# the profiled command runs settle-free: warm-up 1 + timed 3 + clock probe 3 + profile 3 + exchange etc.: count the steps from the number of preproc launches
NS=$(python3 - <<'P'
import csv, glob
n = 0
for f in glob.glob('/tmp/hf/**/*counter_collection.csv', recursive=True):
    n += sum(1 for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('preproc_kernel') and r['Counter_Name'] == 'FETCH_SIZE')
print(n)
P
)
echo "steps in the PMC pass: $NS" >> $OUT/${R}_hbm_per_kernel.txt
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_swin_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' $NS 16 >> $OUT/${R}_hbm_per_kernel.txt; tail -14 $OUT/${R}_hbm_per_kernel.txt
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err; tail -c 400 $OUT/${R}_bench.json; echo
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --fixed-load > $OUT/${R}_bench_fixed_load.json 2>/dev/null
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --fixed-load --roi-size 100,200 > $OUT/${R}_bench_fixed_load_roi_100_200.json 2>/dev/null
python bench.py --batch 64 --steps 30 --no-cpu-baseline --no-fp32-pipe --no-roi-load > $OUT/${R}_bench_b64.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 > $OUT/${R}_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof/*/*kernel_stats.csv $OUT/${R}_kernel_stats.csv; head -8 $OUT/${R}_kernel_stats.csv | cut -c1-140
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/ut -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_util.py '/tmp/ut/**/*counter_collection.csv' > $OUT/${R}_pmc_util.txt; head -14 $OUT/${R}_pmc_util.txt
