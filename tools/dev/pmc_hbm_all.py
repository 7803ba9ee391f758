"""Per kernel: HBM bytes (FETCH_SIZE x2 [gfx950 correction for 16 B/lane streams, MI355X_MICROARCH.md] + WRITE_SIZE, both in KB,
separate --pmc passes) over the kernel's summed duration -> GB/s against the 8 TB/s HBM3E peak."""
import csv, glob, sys, collections
val = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1], recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:56]
        c = row['Counter_Name']
        val[k][c] += float(row['Counter_Value'])
        dur[k][c] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-9
        n[k][c] += 1
rows = []
for k in val:
    if 'FETCH_SIZE' not in val[k] or 'WRITE_SIZE' not in val[k]:
        continue
    t = 0.5 * (dur[k]['FETCH_SIZE'] + dur[k]['WRITE_SIZE'])
    rd, wr = 2 * val[k]['FETCH_SIZE'] * 1024, val[k]['WRITE_SIZE'] * 1024
    rows.append((t, k, n[k]['FETCH_SIZE'], rd, wr))
tot = sum(r[0] for r in rows)
print('%-56s %8s %9s %10s %10s %9s %7s' % ('kernel', 'launches', 'time ms', 'read MB', 'write MB', 'GB/s', '% 8TB/s'))
for t, k, cnt, rd, wr in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    gbs = (rd + wr) / t / 1e9
    print('%-56s %8d %9.2f %10.1f %10.1f %9.0f %6.1f%%' % (k, cnt, t * 1e3, rd / 1e6, wr / 1e6, gbs, 100 * gbs / 8000))
