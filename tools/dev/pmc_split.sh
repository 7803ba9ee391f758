# dev: where do the waves of the split GEMM spend their time (one --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/ps -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle --no-fp32-pipe > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
for f in glob.glob('/tmp/ps/**/*counter_collection.csv', recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:48]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Dispatch_Id'] not in seen:
            seen.add(row['Dispatch_Id']); dur[k] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9
for k, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:6]:
    cyc = c['GRBM_GUI_ACTIVE'] / 8; wc = max(c['SQ_WAVE_CYCLES'], 1)
    print('%-48s %.2f ms clk %.2f GHz mfma_busy %.1f%% | of wave cycles: wait_any %.0f%% wait_inst %.0f%% (lds %.0f%%) active %.0f%% | lds conflict cycles / active lds ~ %.3g' % (
        k, dur[k] * 1e3, cyc / dur[k] / 1e9, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc,
        100 * c['SQ_WAIT_INST_LDS'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc, c['SQ_LDS_BANK_CONFLICT'] / max(cyc, 1)))
PY
