"""Parse rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) into per-launch HBM traffic of a kernel."""
import csv, glob, json, sys
pat = sys.argv[1]
name_key = sys.argv[2] if len(sys.argv) > 2 else 'gemm_kernel<1, 3, 4, 1, 16, 0>'
acc = {}
for f in glob.glob(pat, recursive=True):
    for row in csv.DictReader(open(f)):
        if name_key in row['Kernel_Name']:
            a = acc.setdefault(row['Counter_Name'], [0.0, 0])
            a[0] += float(row['Counter_Value']); a[1] += 1
out = {k: dict(total=v[0], launches=v[1], per_launch=v[0] / max(v[1], 1)) for k, v in acc.items()}
print(json.dumps(out, indent=1))
