# PMC traffic passes of the dominant kernel first (written into profiles/ on the box so that the bench line of the same call
# carries them), then the bench line and the rocprofv3 stats of the same command
R=${ROUND:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hf /tmp/hw /tmp/prof
P="--steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle --no-fp32-pipe"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/hf -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/hw -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' 'gemm_split_kernel<[12], 3, 0>' $GRAFT_REPO_ROOT/nuhtc_amd/csrc/gemm.hip > $OUT/${R}_traffic.json
cp $OUT/${R}_traffic.json $GRAFT_REPO_ROOT/profiles/${R}_traffic.json; cat $OUT/${R}_traffic.json
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_hbm_all.py '/tmp/h[fw]/**/*counter_collection.csv' 24 > $OUT/${R}_hbm_per_kernel.txt
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.err; tail -c 400 $OUT/${R}_bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe > $OUT/${R}_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof/*/*kernel_stats.csv $OUT/${R}_kernel_stats.csv; head -5 $OUT/${R}_kernel_stats.csv
