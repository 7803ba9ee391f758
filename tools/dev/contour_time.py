"""Dev helper: time of the device contour trace (nuhtc_mask_contours) at the bench load (B=16, 256x256 tiles)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import synth, weights, hip
from nuhtc_amd.engine import Engine
eng = Engine(weights.bench_state_dict(0, 5), device=0, max_batch=16)
tiles = eng.to_device(synth.nuclei_tiles(16, 256, start=0))
B = eng.infer_async(tiles, hip.CH_SWAP); eng.check()
for _ in range(3): eng.contours_async(B)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.contours_async(B)
e1.record(); torch.cuda.synchronize()
n = eng.contour_n[:B].cpu().numpy()
print('kept detections traced: %d (overflow %d), mean vertices %.1f, %.1f us per launch' % ((n != 0).sum(), (n < 0).sum(), n[n > 0].mean(), e0.elapsed_time(e1) / 20 * 1e3))
