#!/bin/bash
# rocprofv3 of a short sequential bench: kernel stats, per-dispatch durations of selected kernels, SQ / LDS counters per kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pm1 /tmp/pm2
P="--no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 6 --warmup 2 --in-flight 0 --no-settle"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py $P > $OUT/bench.json 2> $OUT/bench.err
cp /tmp/prof/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 - /tmp/prof/*/*kernel_trace.csv <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name'][:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, n in (('conv3_split_kernel', 16), ('void swin_mlp_kernel', 2), ('window_attn_mfma', 12)):
    for name, v in by.items():
        if name.startswith(k):
            print(name, 'last step, per launch (us):', [round(x, 1) for x in v[-n:]])
P
head -24 $OUT/kernel_stats.csv | cut -c1-140
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pm1 -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pm2 -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_any.py '/tmp/pm1/**/*counter_collection.csv' 10 > $OUT/pmc1.txt; cat $OUT/pmc1.txt
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_any.py '/tmp/pm2/**/*counter_collection.csv' 10 > $OUT/pmc2.txt; cat $OUT/pmc2.txt
