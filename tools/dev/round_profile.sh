set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r01_bench.json 2> gpurun_out/r01_bench.err; tail -c 1500 gpurun_out/r01_bench.json
python bench.py --no-cpu-baseline --in-flight 3 > gpurun_out/r01_bench_pipelined.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats.csv
head -8 $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats.csv
