"""Dev helper: device time of nuhtc_merge_overlap on a synthetic slide's worth of detections."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, torch
from nuhtc_amd import hip, wsi
from test_merge import random_slide
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
rec = random_slide(np.random.default_rng(3), n, 19264 * 2, dup=0.4)
boxes, areas, bits, off = wsi.pack_masks(rec['mask'])
lib = hip.load(); dev = torch.device('cuda', 0)
t = [torch.from_numpy(a).to(dev) for a in (boxes, np.asarray(rec['score'], np.float32), areas, bits.view(np.int32), off)]
keep = torch.zeros(len(areas), dtype=torch.uint8, device=dev)
vp = lambda x: ctypes.c_void_p(x.data_ptr())
def run(mode):
    rc = lib.nuhtc_merge_overlap(0, vp(t[0]), vp(t[1]), vp(t[2]), vp(t[3]), vp(t[4]), len(areas), int(bits.size), mode, 0.05, int(boxes[:, 0].min()),
                                 int(boxes[:, 1].min()), int(boxes[:, 2].max()), int(boxes[:, 3].max()), vp(keep), None)
    assert rc == 0
for mode, name in ((hip.OVERLAP_MASK, 'mask IoU'), (hip.OVERLAP_POLYGON, 'polygon IoU')):
    run(mode); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5): run(mode)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print('%s: %d detections (%.1f MB of mask bits): %d kept, %.2f ms per merge = %.1f M detections/s' % (name, len(areas), bits.nbytes / 1e6, int(keep.sum()), dt * 1e3, len(areas) / dt / 1e6))
