# dev helper: L1/L2 request counters of one kernel of the step (substring in $1), separate --pmc passes
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pk$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pk$i.log 2>&1
done
python3 - "$1" <<'PY'
import csv, glob, collections, sys
key = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(set); dur = {}
for f in glob.glob('/tmp/pk*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if key not in r['Kernel_Name']: continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(key, 'launches', max(len(v) for v in n.values()), 'mean us', sum(dur.values()) / len(dur))
for name, v in sorted(acc.items()):
    print('   %-36s %.5g per launch' % (name, v / len(n[name])))
PY
