import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    d = np.abs(x - y)
    print(f'{k:10s} max abs diff {d.max():.3e}  (|ref| max {np.abs(x).max():.3e})  identical {bool((a[k] == b[k]).all())}')
