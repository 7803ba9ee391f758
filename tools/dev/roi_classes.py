import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import golden_util as G
from nuhtc_amd.engine import Engine
for case in ('small_b2', 'small_wsi_b3', 'full_b1'):
    g = G.load(case)
    eng = Engine(G.seeded_sd(g), device=0, max_batch=len(g['tiles']), tile=g['tiles'].shape[1:3])
    eng.infer_async(eng.to_device(g['tiles']), 0); eng.check()
    print(case, 'rois', int(eng.buffer('roi_total').item()), 'big/mid', eng.buffer('roi_fallback_count').cpu().numpy().tolist())
