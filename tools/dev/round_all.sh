# One GPU call that refreshes every measured artefact of a round (run through gpurun from the repo root):
#   ROUND=r02 bash tools/dev/round_all.sh
# PMC passes first (separate passes, kernel-trace only): HBM bytes per launch of the dominant GEMM -> profiles/$ROUND_traffic.json on
# the box, so that the bench line of the same call carries it; then the bench line (with its streaming figure), the fixed-load bench, the
# rocprofv3 kernel stats of the default bench command and the MFMA-pipe utilisation pass.  Outputs: gpurun_out/$ROUND_*
R=${ROUND:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/ut /tmp/hf /tmp/hw
P="--steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle --no-fp32-pipe --in-flight 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/hf -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/hw -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' 'gemm_split_kernel<[12], 3, 0>' $GRAFT_REPO_ROOT/nuhtc_amd/csrc/gemm.hip > $OUT/${R}_traffic.json; cat $OUT/${R}_traffic.json
cp $OUT/${R}_traffic.json $GRAFT_REPO_ROOT/profiles/${R}_traffic.json
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_hbm_all.py '/tmp/h[fw]/**/*counter_collection.csv' 24 > $OUT/${R}_hbm_per_kernel.txt; head -30 $OUT/${R}_hbm_per_kernel.txt
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err; tail -c 600 $OUT/${R}_bench.json; echo
python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 --fixed-load > $OUT/${R}_bench_fixed_load.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --in-flight 0 > $OUT/${R}_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof/*/*kernel_stats.csv $OUT/${R}_kernel_stats.csv; head -6 $OUT/${R}_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/ut -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_util.py '/tmp/ut/**/*counter_collection.csv' > $OUT/${R}_pmc_util.txt; head -30 $OUT/${R}_pmc_util.txt
