"""Dev check: solo rate of every engine of a process, in creation order (each engine creates two side streams), for a given
GPU_MAX_HW_QUEUES: which stream-to-hardware-queue placements are slow?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict()
tiles = None
OWN = os.environ.get('OWN', '1') == '1'      # run every engine on its own stream (nuhtc_stream) or all on the default stream
def rate(e, n=20):
    with torch.cuda.stream(e.stream if OWN else torch.cuda.current_stream()):
        for _ in range(4): e.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): e.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize(); return 16 * n / (time.perf_counter() - t0)
engines = []
out = []
for i in range(int(os.environ.get('NENG', '10'))):
    e = Engine(sd, device=0, max_batch=16, tile=(256, 256))
    if tiles is None:
        tiles = e.to_device(synth.nuclei_tiles(16, 256))
        for _ in range(30): e.infer_async(tiles, hip.CH_SWAP)
    engines.append(e)
    out.append(round(rate(e)))
print('queues', os.environ.get('GPU_MAX_HW_QUEUES', 'default'), 'solo rate by creation order', out)
print('  again, first to last', [round(rate(e)) for e in engines])
