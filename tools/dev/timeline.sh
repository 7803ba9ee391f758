# dev helper: kernel timeline of the last bench step (start offset, duration, queue) from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roi-load --no-fp32-pipe --no-settle > /tmp/tl.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    rd = csv.DictReader(open(f))
    for r in rd: rows.append(r)
print('columns:', list(rows[0].keys()))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step starts at the last 'preproc' kernel
idx = [i for i, r in enumerate(rows) if 'preproc' in r['Kernel_Name']]
st = idx[-1]
t0 = int(rows[st]['Start_Timestamp'])
prev_end = {}
for r in rows[st:]:
    q = r.get('Queue_Id', '?'); s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print('%9.1f us  +%7.1f us  gap %6.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r['Kernel_Name'][:70]))
PY
