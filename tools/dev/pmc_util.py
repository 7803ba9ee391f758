"""Parse a rocprofv3 --pmc counter_collection CSV: per kernel name, MFMA busy share and effective clock."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for f in glob.glob(sys.argv[1], recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        key = (row['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            dur[k] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9
for k, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:12]:
    if 'GRBM_GUI_ACTIVE' not in c:
        continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    clk = cyc / dur[k] / 1e9
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) if cyc else 0
    print('%-60s time %.2f ms  clock %.2f GHz  MFMA busy %.1f%%  wait_any %.0f%% wait_inst %.0f%%' % (
        k, dur[k] * 1e3, clk, busy * 100, 100 * c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1),
        100 * c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1)))
