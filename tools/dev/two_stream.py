"""Dev experiment: one 16-tile batch as two 8-tile halves on two streams (two engines) vs one engine."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import synth, weights, hip
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict(0, 5)
tiles = torch.from_numpy(synth.nuclei_tiles(16, 256, start=0)).cuda()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
e16 = Engine(sd, device=0, max_batch=16)
print('1 engine  B=16: %.2f ms' % timeit(lambda: e16.infer_async(tiles, hip.CH_SWAP)))
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engs = [Engine(sd, device=0, max_batch=16 // nsplit) for _ in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
parts = tiles.chunk(nsplit)
def split():
    ev = torch.cuda.Event(); ev.record()
    for e, s, p in zip(engs, streams, parts):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            e.infer_async(p, hip.CH_SWAP)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
print('%d engines B=%d each, own streams: %.2f ms' % (nsplit, 16 // nsplit, timeit(split)))
def serial():
    for e, p in zip(engs, parts):
        e.infer_async(p, hip.CH_SWAP)
print('%d engines B=%d each, one stream: %.2f ms' % (nsplit, 16 // nsplit, timeit(serial)))
# two full batches in flight (throughput mode)
k = int(os.environ.get('NENG', 2))
e2 = [e16] + [Engine(sd, device=0, max_batch=16) for _ in range(k - 1)]
def dual():
    ev = torch.cuda.Event(); ev.record()
    for e, s in zip(e2, streams[:k] + [torch.cuda.Stream() for _ in range(max(0, k - len(streams)))]):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            e.infer_async(tiles, hip.CH_SWAP)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
print('%d engines B=16 each, own streams: %.2f ms per 16 tiles' % (k, timeit(dual) / k))
# run-ahead pattern of bench.py: alternate streams without joins
def runahead(n=20):
    for i in range(n):
        with torch.cuda.stream(streams[i % 2]):
            e2[i % 2].infer_async(tiles, hip.CH_SWAP)
for s in streams: s.wait_stream(torch.cuda.current_stream())
runahead(6); torch.cuda.synchronize(); t = time.time(); runahead(20); torch.cuda.synchronize()
print('run-ahead, 2 streams: %.2f ms per 16 tiles' % ((time.time() - t) / 20 * 1e3))
