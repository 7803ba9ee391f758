"""dev (round 6): tools/nuclei_merge.py's two paths on one synthetic GeoJSON: host polygon code (a sample) against the GPU path (whole file)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np
from nuhtc_amd import contours
from test_merge import random_slide
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
rng = np.random.default_rng(3)
rec = random_slide(rng, n, int(40 * n ** 0.5))
feats = []
for (m, x0, y0), s in zip(rec['mask'], rec['score']):
    if m.any():
        c = contours.trace_outer_contour(m)
        feats.append(contours.feature(np.concatenate([c, c[:1]]) + np.array([x0, y0]), int(rng.integers(0, 5)), s, ('T', 'I', 'C', 'D', 'E')))
feats = json.loads(json.dumps(feats))            # as json.load hands them over
contours.merge_features_device(feats[:2000], 0.05)
t0 = time.perf_counter(); dev = contours.merge_features_device(feats, 0.05); t_dev = time.perf_counter() - t0
k = min(len(feats), 8000)
t0 = time.perf_counter(); host = contours.merge_features(feats[:k], 0.05); t_host = time.perf_counter() - t0
print(f'{len(feats)} features -> {len(dev)} kept on the GPU path in {t_dev:.2f} s ({t_dev / len(feats) * 1e6:.1f} us per feature); host polygon path: {k} features in {t_host:.1f} s '
      f'({t_host / k * 1e6:.0f} us per feature; its candidate search grows with the density, not the size)')
