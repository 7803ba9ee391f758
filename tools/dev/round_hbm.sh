# HBM traffic of every kernel of the step: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/hf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/hw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_hbm_all.py '/tmp/h[fw]/**/*counter_collection.csv' 22
