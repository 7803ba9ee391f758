"""Dev check: does an engine created after others were created and closed run as fast as the first one? (GPU_MAX_HW_QUEUES)"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict()
def rate(e, tiles, n=30):
    for _ in range(5): e.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize(); return 16 * n / (time.perf_counter() - t0)
a = Engine(sd, device=0, max_batch=16, tile=(256, 256))
tiles = a.to_device(synth.nuclei_tiles(16, 256))
for _ in range(40): a.infer_async(tiles, hip.CH_SWAP)
print('first engine', round(rate(a, tiles), 1))
others = [Engine(sd, device=0, max_batch=16, tile=(256, 256)) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
for st, e in zip(streams, others):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(5): e.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize()
if os.environ.get('KEEP') != '1':
    for e in others: e.close()
print('first engine again', round(rate(a, tiles), 1))
b = Engine(sd, device=0, max_batch=16, tile=(256, 256))
print('late engine', round(rate(b, tiles), 1))
c = Engine(sd, device=0, max_batch=16, tile=(256, 256))
print('later engine', round(rate(c, tiles), 1))
print('first engine once more', round(rate(a, tiles), 1))
