"""Dev: the four-row form of the 3x3 convolution (two workgroups per CU, NUHTC_CONV_TH4) against the eight-row form in ONE process
(dev build): outputs bit for bit, the convolution tags' time per step, the step alternating between both.   usage: r04_conv_th4.py [rounds]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
NAMES = ['x0', 'x1', 'x2', 'x3', 'sem_feat', 'mask_prob']
def snap():
    B = eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize()
    out = {'counts': eng.counts[:B].clone(), 'boxes': eng.boxes[:B].clone(), 'labels': eng.labels[:B].clone(), 'masks': eng.masks[:B].clone()}
    for n in NAMES:
        try: out[n] = eng.buffer(n)
        except Exception as ex: print('no buffer', n, ex)
    return out
for _ in range(20): eng.infer_async(tiles, hip.CH_SWAP)
hip.dev_knob('CONV_TH4', 0); a = snap()
hip.dev_knob('CONV_TH4', 1); b = snap()
for k in a:
    if k in b:
        same = torch.equal(a[k], b[k])
        extra = '' if same or not a[k].is_floating_point() else f' max abs diff {float((a[k] - b[k]).abs().max()):.3e}'
        print(f'{k}: {"identical" if same else "DIFFERENT"}{extra}')
t = {0: [], 1: []}
for r in range(rounds):
    for v in ((0, 1) if r % 2 == 0 else (1, 0)):
        hip.dev_knob('CONV_TH4', v)
        for _ in range(2): eng.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eng.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize(); t[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v in (0, 1):
    x = np.array(t[v]); print(f'CONV_TH4={v}: {x.mean():.3f} ms per step (min {x.min():.3f}) -> {16e3 / x.mean():.1f} tiles/s sequential')
for v in (0, 1):
    hip.dev_knob('CONV_TH4', v)
    hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    g = {k: round(x['ms'] / 5, 4) for k, x in sorted(p.items()) if 'conv3' in k}
    print(f'  CONV_TH4={v}:', g, 'sum', round(sum(g.values()), 4))
