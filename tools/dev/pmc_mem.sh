# dev helper: memory-path counters per GEMM launch shape (default: tools/dev/gemm_big.py; PMC_CMD/PMC_ARGS choose
# another script, e.g. PMC_CMD=bench.py PMC_ARGS='--steps 2 --warmup 1 --no-cpu-baseline'); separate passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm$i -- python3 $GRAFT_REPO_ROOT/${PMC_CMD:-tools/dev/gemm_big.py} ${PMC_ARGS} > /tmp/pm$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob('/tmp/pm*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, c in acc.items():
    if 'gemm' not in k or sum(len(n[(k, nm)]) for nm in c) < 8: continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-36s %.4g per launch' % (name, v / len(n[(k, name)])))
PY
