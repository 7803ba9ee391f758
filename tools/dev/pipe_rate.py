"""Dev: rate of the EnginePipeline loop (four batches in flight) with device-resident / host tiles, with / without the export path
and the host-side unpack: where does the slide loop lose against bench.py's `value`?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights, wsi
from nuhtc_amd.pipeline import EnginePipeline
sd = weights.bench_state_dict()
DEPTH = int(os.environ.get('DEPTH', '4'))
pipe = EnginePipeline(sd, device=0, depth=DEPTH, max_batch=16, tile=(256, 256))
host = [synth.nuclei_tiles(16, 256, start=16 * i) for i in range(8)]
dev = [pipe.engines[0].to_device(h) for h in host]
def loop(n, src, export, unpack):
    parts = []
    t0 = None
    for i in range(n + 8):
        if i == 8:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if pipe.full():
            eng, B, st, tag = pipe.collect()
            if unpack:
                with torch.cuda.stream(st):
                    g = eng.export_read()
                    if g is not None and g['n']:
                        wsi._unpack_packed(eng, g, 0, np.zeros((B, 2), np.int64), parts)
        pipe.submit(src[i % 8], hip.CH_SWAP, tag=i, export=export)
    for _ in pipe.drain(): pass
    torch.cuda.synchronize()
    return 16 * n / (time.perf_counter() - t0)
cases = (('device tiles', dev, False, False), ('device tiles + export', dev, True, False), ('device tiles + export + unpack', dev, True, True),
         ('host tiles + export + unpack', host, True, True), ('host tiles', host, False, False), ('device tiles', dev, False, False))
if os.environ.get('SHORT'):
    cases = (cases[0], cases[3], cases[0], cases[3])
print('queues', os.environ.get('GPU_MAX_HW_QUEUES', 'default'), 'depth', DEPTH)
for name, src, export, unpack in cases:
    print(f'   {name:34s} {loop(120, src, export, unpack):7.0f} tiles/s')
