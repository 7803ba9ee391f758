"""Summarise /tmp/conv_stamps.txt (-DNUHTC_CONV_STAMPS): per-wave phase cycles of conv3_split_kernel, per tile."""
import sys
import numpy as np
f = sys.argv[1] if len(sys.argv) > 1 else '/tmp/conv_stamps.txt'
print(open(f).readline().strip())
a = np.loadtxt(f)
a = a[a[:, 6] > 0]
nt = a[:, 6]
names = ['split', 'main', 'epilogue', 'barriers']
for half, sel in (('waves 0-3', a[:, 1] < 4), ('waves 4-7', a[:, 1] >= 4)):
    v = a[sel]
    per = v[:, 2:6] / v[:, 6:7]
    print(half, ' '.join(f'{n} {m:.0f}' for n, m in zip(names, per.mean(0))), '| per tile total', f'{(v[:, 7] / v[:, 6]).mean():.0f}', 'tiles per workgroup', f'{v[:, 6].mean():.1f}')
print('kernel span (cycles):', (a[:, 8] + a[:, 7]).max() - a[:, 8].min())
