"""Dev: cProfile of the slide loop (wsi.infer_tiles over a synthetic 40x40 slide)."""
import cProfile, pstats, os, sys, time, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import apis, synth, wsi, weights
G = 40
band, y0 = synth.nuclei_canvas_parallel(G, rows=(0, G), workers=16)
tiles = synth.CanvasTiles(band, y0, G, 0, G * G)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    model = apis.init_detector(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'), None, 'cuda:0')
model.state_dict = weights.bench_state_dict()
wsi.infer_tiles(model, tiles[0:64], tiles.coords[:64], 16)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
rec = wsi.infer_tiles(model, tiles, tiles.coords, 16)
pr.disable()
print(f'{G * G / (time.perf_counter() - t0):.1f} tiles/s under the profiler')
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
