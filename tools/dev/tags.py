"""dev: kernel tags of the current build whose name contains one of the given substrings: ms per step, one batch at a time (profile events).
usage: tags.py conv3 window_attn ...   [TAG_RUNS=2]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
pats = sys.argv[1:] or ['']
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
for _ in range(30): eng.infer_async(tiles, hip.CH_SWAP)
for r in range(int(os.environ.get('TAG_RUNS', 2))):
    torch.cuda.synchronize(); hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    g = {k: round(x['ms'] / 5, 4) for k, x in sorted(p.items()) if any(q in k for q in pats)}
    print(g, 'sum', round(sum(g.values()), 4))
