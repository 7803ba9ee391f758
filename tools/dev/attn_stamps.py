"""Dev helper: summarise /tmp/attn_stamps_<call>.txt written by a -DNUHTC_ATTN_STAMPS build of swin.hip (per-wave s_memtime
stamps of window_attn_mfma_kernel: start, K/V landed, then per query tile: Q/bias landed, S done, softmax done, PV done; end)."""
import sys, glob, numpy as np
names = ['K,V loads', 'ti0 Q/bias', 'ti0 S mfma', 'ti0 softmax', 'ti0 PV mfma', 'ti1 Q/bias (+store ti0)', 'ti1 S mfma', 'ti1 softmax', 'ti1 PV mfma', 'store ti1']
for f in sorted(glob.glob('/tmp/attn_stamps_*.txt')):
    hdr = open(f).readline().split()
    d = np.loadtxt(f, skiprows=1, dtype=np.float64)
    d = d[d[:, 0] > 0]
    print(f, 'nPairs C nH mask =', hdr, 'waves sampled', len(d), 'span', int(d[:, 10].max() - d[:, 0].min()), 'cycles')
    for i in range(10):
        x = d[:, i + 1] - d[:, i]
        print('   %-26s mean %7.0f  p50 %7.0f  p90 %7.0f' % (names[i], x.mean(), np.percentile(x, 50), np.percentile(x, 90)))
    tot = d[:, 10] - d[:, 0]
    print('   %-26s mean %7.0f  p50 %7.0f' % ('wave lifetime', tot.mean(), np.percentile(tot, 50)))
