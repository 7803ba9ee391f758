cd /tmp && export TMPDIR=/tmp
SHAPES="${PMC_SHAPES:-262144x384x96}"
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" "TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum TCC_IB_STALL_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/wr$i -- python3 $GRAFT_REPO_ROOT/tools/dev/gemm_iso.py $SHAPES > /tmp/wr$i.log 2>&1 || tail -3 /tmp/wr$i.log
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob('/tmp/wr*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, c in acc.items():
    if 'gemm' not in k: continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-44s %.5g per launch' % (name, v / len(n[(k, name)])))
PY
