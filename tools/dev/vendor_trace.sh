cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/vt -- python3 $GRAFT_REPO_ROOT/tools/dev/gemm_vendor.py 262144x384x96 65536x768x192 16384x1536x384 16384x3072x3072 > /tmp/vt.log 2>&1
python3 - <<'PY'
import csv, glob, collections
seen = collections.OrderedDict()
for f in glob.glob('/tmp/vt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm_kernel' in k or 'elementwise' in k or 'distribution' in k: continue
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        key = (k, r.get('Grid_Size'), r.get('Workgroup_Size'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'))
        seen.setdefault(key, []).append(d)
for k, v in seen.items():
    print(k[0][:400]); print('    grid', k[1], 'wg', k[2], 'lds', k[3], 'vgpr', k[4], 'agpr', k[5], 'sgpr', k[6], 'n', len(v), 'avg us %.1f' % (sum(v) / len(v)))
PY
