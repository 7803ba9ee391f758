"""dev (round 6): launch time per full kernel tag under the FIXED load (1064 given RoIs, 64 detections per tile: counts do not follow the results, so
result-altering probe builds still time the same work).   python tools/dev/r06_tags_fixed.py [pattern ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
pats = sys.argv[1:]
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256), bind_host=True)
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
rois = torch.from_numpy(synth.fixed_load_rois(16)).to(tiles.device)
step = lambda: eng.infer_fixed_load_async(tiles, rois, 64, hip.CH_SWAP)
for _ in range(40): step()
torch.cuda.synchronize(); hip.profile_enable(True)
for _ in range(6): step()
p = hip.profile_read()
for k, x in sorted(p.items(), key=lambda kv: -kv[1]['ms']):
    if not pats or any(q in k for q in pats):
        print(f"{k:50s} {x['launches'] // 6:3d} launches {x['ms'] / 6:7.3f} ms  {x['ms'] / max(1, x['launches']) * 1e3:7.1f} us each")
