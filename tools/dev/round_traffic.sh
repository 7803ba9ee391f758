# HBM traffic of the dominant GEMM kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_traffic.py '/tmp/t[fw]/**/*counter_collection.csv' 'gemm_kernel<1, 3, 4, 1, 16, 0>'
