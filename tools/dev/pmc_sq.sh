# dev helper: wave-scheduler counters of isolated GEMM shapes (tools/dev/gemm_iso.py MxNxK ...), one --pmc pass per set
cd /tmp && export TMPDIR=/tmp
SHAPES="${PMC_SHAPES:-262144x384x96 16384x3072x3072}"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq$i -- python3 $GRAFT_REPO_ROOT/tools/dev/gemm_iso.py $SHAPES > /tmp/sq$i.log 2>&1 || tail -3 /tmp/sq$i.log
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob('/tmp/sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, c in acc.items():
    if 'gemm' not in k: continue
    print(k)
    for name, v in sorted(c.items()):
        print('   %-36s %.5g per launch' % (name, v / len(n[(k, name)])))
PY
