"""Dev: run one batch and save a few stage tensors (A/B of kernel variants across processes: knobs come from the environment).
    python tools/dev/dump_stage.py out.npz [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = weights.bench_state_dict()
eng = Engine(sd, device=0, max_batch=B, tile=(256, 256))
tiles = synth.nuclei_tiles(B, 256, start=0)
eng.infer_async(eng.to_device(tiles), hip.CH_SWAP)
eng.check()
out = {k: eng.buffer(k)[:B].cpu().numpy() for k in ('c0', 'c3', 'x0', 'x1', 'x2', 'x3', 'rpn0', 'rpn3', 'sem_feat', 'sem_pred')}
out['counts'] = eng.counts[:B].cpu().numpy()
out['boxes'] = eng.boxes[:B].cpu().numpy()
out['mask_prob'] = eng.buffer('mask_prob').cpu().numpy()
np.savez(sys.argv[1], **out)
print('saved', sys.argv[1], 'counts', out['counts'])
