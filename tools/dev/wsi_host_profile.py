"""Dev: where does the slide loop spend its time?  Runs wsi.infer_tiles over a synthetic 40x40 slide with (a) the real host unpack,
(b) the unpack replaced by the bare export_read."""
import os, sys, time, numpy as np, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import apis, synth, wsi
G = 40
band, y0 = synth.nuclei_canvas_parallel(G, rows=(0, G), workers=16)
tiles = synth.CanvasTiles(band, y0, G, 0, G * G)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = apis.init_detector(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'), None, 'cuda:0')
from nuhtc_amd import weights
model.state_dict = weights.bench_state_dict()
wsi.infer_tiles(model, tiles[0:64], tiles.coords[:64], 16)
for mode in ('real', 'no-unpack', 'real'):
    real = wsi._unpack
    if mode == 'no-unpack':
        wsi._unpack = lambda eng, B, i0, coords, P, rec, exported=False: eng.export_read()
    t0 = time.perf_counter()
    rec = wsi.infer_tiles(model, tiles, tiles.coords, 16)
    dt = time.perf_counter() - t0
    wsi._unpack = real
    print(mode, f'{G * G / dt:.1f} tiles/s', len(rec['tile']), 'records')
