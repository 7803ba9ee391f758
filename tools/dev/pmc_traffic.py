"""Parse rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) into the per-launch HBM traffic of a kernel.

    pmc_traffic.py '<glob of counter_collection.csv>' '<kernel name regex>' [path of the kernel's source file]

Prints the JSON bench.py reads from profiles/rNN_traffic.json: FETCH_SIZE doubled (gfx950 tallies wide coalesced reads at
half their bytes, MI355X_MICROARCH.md HBM section), WRITE_SIZE as is, and the sha1 of the source the numbers belong to."""
import csv, glob, hashlib, json, re, sys
pat = sys.argv[1]
name_key = sys.argv[2] if len(sys.argv) > 2 else 'gemm_kernel<1, 3, 4, 1, 16, 0>'
acc = {}
for f in glob.glob(pat, recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(name_key, row['Kernel_Name']) and row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            a = acc.setdefault(row['Counter_Name'], [0.0, 0])
            a[0] += float(row['Counter_Value']); a[1] += 1
fetch = acc.get('FETCH_SIZE', [0.0, 0]); write = acc.get('WRITE_SIZE', [0.0, 0])
fk, wk = fetch[0] / max(fetch[1], 1), write[0] / max(write[1], 1)       # KB per launch (rocprofv3 reports these counters in KB)
out = {'kernel': name_key, 'command': 'tools/dev/round_all.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle ; same with --pmc WRITE_SIZE (separate passes)',
       'launches_counted': fetch[1], 'FETCH_SIZE_per_launch_KB': round(fk, 1), 'WRITE_SIZE_per_launch_KB': round(wk, 1),
       'gfx950_correction': 'FETCH_SIZE doubled (16 B/lane coalesced dwordx4 streams are tallied at half their bytes, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is',
       'hbm_bytes_per_launch': int((2 * fk + wk) * 1024)}
if len(sys.argv) > 3:
    out['gemm_hip_sha1'] = hashlib.sha1(open(sys.argv[3], 'rb').read()).hexdigest()
print(json.dumps(out, indent=1))
