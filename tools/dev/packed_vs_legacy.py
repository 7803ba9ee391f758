import os, sys, warnings, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import apis, synth, wsi, weights
G = 6
band, y0 = synth.nuclei_canvas_parallel(G, rows=(0, G), workers=4)
tiles = synth.CanvasTiles(band, y0, G, 0, G * G)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = apis.init_detector('configs/nuhtc/htc_lite_swin_pannuke_infer.py', None, 'cuda:0', max_batch=8)
model.state_dict = weights.bench_state_dict()
rec = wsi.infer_tiles(model, tiles, tiles.coords, 8)
print(type(rec['mask']).__name__, len(rec['tile']))
# legacy path: force per-detection unpack
real = wsi._unpack_packed
def legacy_finish_patch():
    pass
import torch
rec2 = dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
pipe = model.pipeline(tiles.shape[1:3], 4)
from nuhtc_amd import hip
pend = []
for i in range(0, len(tiles), 8):
    if pipe.full():
        eng, B, stream, i0 = pipe.collect()
        with torch.cuda.stream(stream):
            wsi._unpack(eng, B, i0, tiles.coords, 256, rec2, exported=False)
    pipe.submit(tiles[i:i + 8], hip.CH_SWAP, tag=i, export=True)
while pipe.pending:
    eng, B, stream, i0 = pipe.collect()
    with torch.cuda.stream(stream):
        wsi._unpack(eng, B, i0, tiles.coords, 256, rec2, exported=False)
assert rec['tile'] == rec2['tile'] and rec['score'] == rec2['score'] and rec['label'] == rec2['label'], 'scalars differ'
assert all(np.array_equal(a, b) for a, b in zip(rec['box'], rec2['box']))
assert all(np.array_equal(a, b) for a, b in zip(rec['ring'], rec2['ring'])), 'rings differ'
for (m, x0, y0), (m2, x2, y2) in zip(rec['mask'], rec2['mask']):
    assert (x0, y0) == (x2, y2) and np.array_equal(m, m2)
pk = wsi.pack_records(rec); pk2 = wsi.pack_records(rec2)
for a, b in zip(pk, pk2):
    assert a.shape == b.shape and bool((a == b).all()), (a.shape, b.shape)
keep = list(range(0, len(rec['tile']), 3))
for a, b in zip(wsi.pack_records(rec, keep), wsi.pack_records(rec2, keep)):
    assert a.shape == b.shape and bool((a == b).all())
k1 = wsi.merge_overlap(rec, 0.05); k2 = wsi.merge_overlap(rec2, 0.05)
assert np.array_equal(k1, k2)
print('packed path == per-detection path:', len(rec['tile']), 'records,', len(k1), 'kept after merge')
