"""Dev helper: soak run -- N steps of the full path on one batch; outputs must stay bit-identical, device memory flat."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from nuhtc_amd import synth, weights, hip
from nuhtc_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
eng = Engine(weights.bench_state_dict(0), device=0, max_batch=16, tile=(256, 256))
tiles = eng.to_device(synth.nuclei_tiles(16, 256, start=0))
def digest():
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for t in (eng.counts, eng.boxes, eng.labels, eng.keep, eng.masks):
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()
eng.infer_async(tiles, hip.CH_SWAP); ref = digest()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
for i in range(1, n + 1):
    eng.infer_async(tiles, hip.CH_SWAP)
    if i % 500 == 0:
        d = digest(); eng.check()
        print(i, 'identical' if d == ref else 'DIFFERENT', 'free MB delta', (torch.cuda.mem_get_info()[0] - free0) >> 20, '%.1f tiles/s' % (16 * i / (time.time() - t0)))
