"""Dev: RoI size classes at the bench load (big / mid / giant counts and the sides of the big boxes)."""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'); sys.path.insert(0, R)
import numpy as np, torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict()
e = Engine(sd, device=0, max_batch=16, tile=(256, 256))
tiles = e.to_device(synth.nuclei_tiles(16, 256))
e.infer_async(tiles, hip.CH_SWAP); e.check(); torch.cuda.synchronize()
cnt = e.buffer('roi_fallback_count').cpu().numpy().tolist()
n = int(e.buffer('roi_total').item())
rois = e.buffer('rois').cpu().numpy()[:n]
w = rois[:, 3] - rois[:, 1]; h = rois[:, 4] - rois[:, 2]
side = np.maximum(w, h)
print('rois (last stage)', n, 'big/mid/giant', cnt)
print('max side histogram (px):', np.histogram(side, bins=[0, 20, 40, 56, 80, 112, 160, 256, 384, 520])[0].tolist())
