"""Sum rocprofv3 --pmc counters per kernel: python pmc_any.py '<glob of counter_collection.csv>' [top N]"""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1], recursive=True)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            n[k] += 1
names = sorted({c for v in acc.values() for c in v})
order = sorted(acc, key=lambda k: -acc[k].get('GRBM_GUI_ACTIVE', 0))[:top]
for k in order:
    v = acc[k]
    print(k, 'dispatches', n[k])
    print('    ' + '  '.join(f'{c}={v[c]:.4g}' for c in names))
    wc = v.get('SQ_WAVE_CYCLES', 0)
    if wc:
        s = []
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_MISC'):
            if c in v:
                s.append(f'{c[3:]} {100 * v[c] / wc:.0f}%')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE'):
            s.append(f"MFMA busy {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.0f}%")     # 1024 SIMDs; GUI_ACTIVE summed over 8 XCDs
        if 'SQ_LDS_BANK_CONFLICT' in v and v.get('SQ_LDS_IDX_ACTIVE'):
            s.append(f"LDS conflict cycles {100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.0f}% of LDS active")
        print('    ' + ', '.join(s))
