"""Dev helper: end-to-end rate of nuhtc_amd.wsi.infer_tiles (engine + device contours + host unpacking) on synthetic tiles."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from nuhtc_amd import synth, weights, wsi
from nuhtc_amd.apis import init_detector
ck = '/tmp/w.pth'
torch.save(dict(state_dict=weights.bench_state_dict(0)), ck)
model = init_detector(os.path.join(os.path.dirname(__file__), '..', '..', 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'), ck, device='cuda:0', max_batch=16)
model.opts.update(margin=2, min_area=10, mask_nms_thr=0.05)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tiles = synth.nuclei_tiles(n, 256, start=0)
coords = np.stack([np.arange(n) % 16 * 192, np.arange(n) // 16 * 192], 1)
wsi.infer_tiles(model, tiles[:48], coords[:48], 16)
t = time.time(); rec = wsi.infer_tiles(model, tiles, coords, 16); dt = time.time() - t
print('%d tiles, %d kept detections: %.1f tiles/s end to end' % (n, len(rec['score']), n / dt))
pr = cProfile.Profile(); pr.enable(); wsi.infer_tiles(model, tiles[:96], coords[:96], 16); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
