#!/bin/bash
# timeline of the last sequential step: kernel, start (us from the step's first kernel), duration, queue
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
P="--no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 4 --warmup 2 --in-flight 0 --no-settle"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py $P > $OUT/bench.json 2> $OUT/bench.err
python3 - /tmp/prof/*/*kernel_trace.csv > $OUT/timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step that is followed by nothing but the bench's tail: find the last preproc_kernel that has a complete step after it
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('preproc_kernel')]
# use the 4th from last preproc (timed steps; later ones belong to the clock probe / profiling phases too, any is fine)
i0 = idx[-3]; i1 = idx[-2]
t0 = int(rows[i0]['Start_Timestamp'])
qs = {}
for r in rows[i0:i1]:
    q = qs.setdefault(r['Queue_Id'], len(qs))
    s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{s:9.1f} {d:8.1f}  q{q}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X','')} wg {r.get('Workgroup_Size_X','')}")
print('step span us', (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
P
tail -150 $OUT/timeline.txt
