"""Dev check: solo rate of ONE engine created after K dummy streams (its side streams are streams K+1 and K+2 of the process), for
the given GPU_MAX_HW_QUEUES: which placements relative to the caller's (null) stream are slow?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict()
K = int(sys.argv[1])
torch.cuda.init()
dummies = [torch.cuda.Stream() for _ in range(K)]
for d in dummies:
    with torch.cuda.stream(d): torch.zeros(1, device='cuda')
torch.cuda.synchronize()
e = Engine(sd, device=0, max_batch=16, tile=(256, 256))
tiles = e.to_device(synth.nuclei_tiles(16, 256))
for _ in range(30): e.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): e.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize()
print('queues', os.environ.get('GPU_MAX_HW_QUEUES', 'default'), 'dummies', K, 'rate', round(16 * 30 / (time.perf_counter() - t0)))
