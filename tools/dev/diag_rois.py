import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict()
eng = Engine(sd, device=0, max_batch=16, tile=(256,256))
tiles = eng.to_device(synth.nuclei_tiles(16,256))
eng.enable_token_dump()
eng.infer_async(tiles, 1); eng.check()
R = int(eng.buffer('roi_total').item())
print('R', R, 'fallback (last stage)', int(eng.buffer('roi_fallback_count').item()))
for k in range(3):
    r = eng.buffer(f'rois_stage{k}')[:R].cpu().numpy()
    w = r[:,3]-r[:,1]; h = r[:,4]-r[:,2]
    print(k, 'w quantiles', np.quantile(w,[0,.1,.25,.5,.75,.9,.99,1]).round(1), 'frac w<=24', (w<=24).mean(), 'max(w,h)<=28', (np.maximum(w,h)<=28).mean())
