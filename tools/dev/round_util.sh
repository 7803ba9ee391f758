# MFMA-pipe utilisation and effective clock of the dominant kernels: one --pmc pass (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/ut -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_util.py '/tmp/ut/**/*counter_collection.csv'
