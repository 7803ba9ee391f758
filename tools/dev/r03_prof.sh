#!/bin/bash
# rocprofv3 kernel trace of a short sequential bench: per-kernel stats + per-dispatch durations of selected kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
rm -rf /tmp/prof && mkdir -p /tmp/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python3 bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 10 --warmup 2 --in-flight 0 --no-settle > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3p/kernel_stats.csv
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in ('conv3_split_kernel', 'void swin_mlp_kernel', 'window_attn_mfma'):
    for name, v in by.items():
        if name.startswith(k):
            v2 = v[len(v)//2:]     # second half of the run
            n = {'conv3_split_kernel': 16, 'void swin_mlp_kernel': 2, 'window_attn_mfma': 12}[k]
            print(name, 'per-step pattern (us):', [round(x, 1) for x in v2[-n:]])
P
head -30 gpurun_out/r3p/kernel_stats.csv | cut -c1-150
