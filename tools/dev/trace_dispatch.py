"""Dev helper: aggregate a rocprofv3 --kernel-trace CSV per (kernel, grid size): launches, avg us, total ms."""
import csv, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:60]
        g = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r.get('Grid_Size_Z', 1)))
        k = (name, g)
        agg[k][0] += 1
        agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print('%-62s grid=%-14s n=%5d avg=%9.1f us total=%8.2f ms (%4.1f%%)' % (k[0], k[1], v[0], v[1] / v[0], v[1] / 1e3, 100 * v[1] / tot))
