"""Dev helper: summarise /tmp/stamps.txt, the per-wave s_memtime stamps one GEMM launch writes when libnuhtc_hip.so is built
with -DNUHTC_GEMM_STAMPS (phase durations in shader cycles, barrier / staging-load waits inside the k-loop)."""
import numpy as np, collections
d = np.loadtxt('/tmp/stamps.txt', dtype=np.uint64)
b, w, t0, t1, t2, t3, t4, hw = [d[:, i].astype(np.int64) for i in range(8)]
bar, tile0 = [d[:, i].astype(np.int64) for i in (8, 9)]
ok = t0 > 0
base = t0[ok].min()
print('waves', ok.sum(), 'span cycles(100MHz ticks?)', (t4[ok].max() - base))
for name, x in (('prologue', t1 - t0), ('mainloop', t2 - t1), ('epilogue', t3 - t2), ('drain', t4 - t3), ('total', t4 - t0), ('loop:barrier wait', bar), ('loop:load wait', tile0)):
    x = x[ok]; print('%-22s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (name, x.mean(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90)))
import os
if os.environ.get('STAMPS_RT'):
    life = (t4 - t0)[ok]; rt = tile0[ok]
    print('s_memtime ticks per 100 MHz tick: mean %.2f -> %.2f GHz' % ((life / np.maximum(rt, 1)).mean(), (life.sum() / rt.sum()) / 10))
# per-XCD spans: every XCD has its own counter base, so cluster the start stamps
o = np.argsort(t0[ok]); ts = t0[ok][o]; te = t4[ok][o]
cuts = np.nonzero(np.diff(ts) > 400000)[0] + 1
spans = []
for a, b_ in zip(np.r_[0, cuts], np.r_[cuts, len(ts)]):
    spans.append((te[a:b_].max() - ts[a], np.percentile(ts[a:b_] - ts[a], [50, 90, 100]).astype(int).tolist(), b_ - a))
print('per-XCD span (ticks), start offsets p50/p90/max, waves:', spans[:8])
# per CU timeline for one CU: (se, sh?, cu) from HW_ID
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3; wid = hw & 0xF
xcc = (hw >> 20) & 0xF
key = collections.Counter(zip(xcc[ok].tolist(), se[ok].tolist(), cu[ok].tolist()))
print('distinct (xcc,se,cu):', len(key), 'waves per key min/max', min(key.values()), max(key.values()))
k0 = sorted(key)[0]
sel = ok & (xcc == k0[0]) & (se == k0[1]) & (cu == k0[2]) & (simd == 0)
order = np.argsort(t0[sel])
print('timeline of SIMD0 of', k0, '(start, first-mfma, loop-end, epi-end, drained, wave slot):')
for i in order[:24]:
    j = np.nonzero(sel)[0][i]
    print('  blk %5d w%d  %8d %8d %8d %8d %8d  slot %d' % (b[j], w[j], t0[j] - base, t1[j] - base, t2[j] - base, t3[j] - base, t4[j] - base, wid[j]))
