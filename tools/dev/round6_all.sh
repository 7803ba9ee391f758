# One GPU call that refreshes the measured artefacts of round 6 (run through gpurun from the repo root, default build):
#   bash tools/dev/round6_all.sh [quick]        ("quick" skips the 7-minute full-protocol CPU baseline and the slide-level bench)
R=r06
OUT=$GRAFT_REPO_ROOT/gpurun_out/round6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/ut /tmp/hf /tmp/hw /tmp/lds
P="--steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle --no-fp32-pipe --in-flight 0 --no-force-collective"   # (no one-rank RCCL communicator under counter collection: ADVICE r5)
# ---- PMC passes (separate passes, kernel trace only): HBM bytes, LDS conflicts, MFMA utilisation
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/hf -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/hw -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' 'gemm_split_kernel<[12], 3, [02]>' $GRAFT_REPO_ROOT/nuhtc_amd/csrc/gemm.hip > $OUT/${R}_traffic.json; cat $OUT/${R}_traffic.json
cp $OUT/${R}_traffic.json $GRAFT_REPO_ROOT/profiles/${R}_traffic.json          # the bench line below then carries `roofline.traffic`
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_hbm_all.py '/tmp/h[fw]/**/*counter_collection.csv' 30 > $OUT/${R}_hbm_per_kernel.txt
NS=$(python3 - <<'P'
import csv, glob
n = 0
for f in glob.glob('/tmp/hf/**/*counter_collection.csv', recursive=True):
    n += sum(1 for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('patch_embed') and r['Counter_Name'] == 'FETCH_SIZE')
print(n)
P
)
echo "steps in the PMC pass: $NS" >> $OUT/${R}_hbm_per_kernel.txt
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_swin_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' $NS 16 >> $OUT/${R}_hbm_per_kernel.txt; tail -14 $OUT/${R}_hbm_per_kernel.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/lds -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_any.py '/tmp/lds/**/*counter_collection.csv' > $OUT/${R}_pmc_lds.txt; head -12 $OUT/${R}_pmc_lds.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/ut -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_util.py '/tmp/ut/**/*counter_collection.csv' > $OUT/${R}_pmc_util.txt; head -14 $OUT/${R}_pmc_util.txt
# ---- bench lines (the default command with its 10 Hz power log; three more processes back to back: the state of the dense launches
# differs by process, the power / clock columns go with each)
cd $GRAFT_REPO_ROOT
python bench.py --power-csv $OUT/${R}_power_bench.csv > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err; tail -c 300 $OUT/${R}_bench.json; echo
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 300 --power-csv $OUT/${R}_power_run$i.csv > $OUT/${R}_bench_run$i.json 2>/dev/null; done
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --fixed-load > $OUT/${R}_bench_fixed_load.json 2>/dev/null
python bench.py --batch 64 --steps 30 --no-cpu-baseline --no-fp32-pipe --no-roi-load > $OUT/${R}_bench_b64.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 > $OUT/${R}_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof/*/*kernel_stats.csv $OUT/${R}_kernel_stats.csv; head -8 $OUT/${R}_kernel_stats.csv | cut -c1-140
cd $GRAFT_REPO_ROOT
if [ "$1" != "quick" ]; then
  timeout 900 python tools/bench_wsi.py > $OUT/${R}_bench_wsi.json 2> $OUT/${R}_bench_wsi.err; tail -c 400 $OUT/${R}_bench_wsi.json; echo
  timeout 900 python tools/bench_wsi.py --grid 40 --svs jpeg > $OUT/${R}_bench_wsi_svs.json 2> $OUT/${R}_bench_wsi_svs.err
  timeout 1500 python bench.py --cpu-full --no-roi-load --no-fp32-pipe --steps 20 > $OUT/${R}_cpu_baseline_full.json 2> $OUT/${R}_cpu_baseline_full.err; tail -c 700 $OUT/${R}_cpu_baseline_full.json; echo
fi
