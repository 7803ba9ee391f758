"""Summarise /tmp/mlp_stamps.txt (a -DNUHTC_MLP_STAMPS build): per-wave phase cycles of swin_mlp_kernel."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1] if len(sys.argv) > 1 else '/tmp/mlp_stamps.txt')
names = ['prologue', 'gemm1', 'act', 'gemm2', 'stage', 'barrier', 'epilogue']
for half, sel in (('waves 0-3', a[:, 1] < 4), ('waves 4-7', a[:, 1] >= 4)):
    v = a[sel][:, 2:9]
    print(half, ' '.join(f'{n} {m:.0f}' for n, m in zip(names, v.mean(0))), 'total', f'{v.sum(1).mean():.0f}')
t0 = a[:, 9]
print('block start spread (cycles):', t0.max() - t0.min(), 'blocks', int(a[:, 0].max()) + 1)
