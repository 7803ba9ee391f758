"""ORACLE-side tool (test infrastructure): one-command parity check of the HIP engine against the CPU restatement on a REAL
checkpoint (`models/pannuke.pth` is not distributed with the reference, so parity with it can only be shown
where the file is available — this is that check).

    python -m oracle.check_checkpoint --checkpoint models/pannuke.pth \
        [--config configs/nuhtc/htc_lite_swin_pannuke_infer.py] [--tiles 16] [--images DIR] [--size 256] [--mode wsi|file]

Loads the checkpoint the way `init_detector` does (nuhtc/apis/inference.py:44, non-strict, EMA / optimizer entries
ignored), runs the same tiles through oracle/model.py on the host cores and through libnuhtc_hip.so on cuda:0, and prints
one line per tile: instances of each side, matched one to one (same class, box IoU >= 0.999, |score diff| < 1e-3), mask
pixels that differ and how far their pasted probability is from 0.5, every tolerated disagreement with the threshold that
explains it (tests/parity_util.py).  Exit code 0 iff every tile passes.  Tiles: PNG/JPG files of --images (first --tiles
of them, all of one size) or synthetic nuclei tiles.  --mode wsi = ndarray input of tools/infer_wsi.py (channels swapped
by the BGR pipeline), file = tools/infer.py (true RGB)."""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--checkpoint', required=True)
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'))
    ap.add_argument('--tiles', type=int, default=16)
    ap.add_argument('--images', default=None, help='directory of tile images (default: synthetic nuclei tiles)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--mode', default='wsi', choices=['wsi', 'file'])
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    import torch
    import parity_util as P
    from nuhtc_amd import synth, weights
    from nuhtc_amd.config import Config, engine_options, patch_config
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    cfg = patch_config(Config.fromfile(a.config))
    opts = engine_options(cfg)
    nc = opts.pop('num_classes')
    sd = weights.load_checkpoint(a.checkpoint, nc)
    if a.images:
        from PIL import Image
        files = sorted(f for f in glob.glob(os.path.join(a.images, '*')) if f.lower().endswith(('.png', '.jpg', '.jpeg', '.tif', '.tiff')))[:a.tiles]
        tiles = np.stack([np.asarray(Image.open(f).convert('RGB')) for f in files])
    else:
        tiles = synth.nuclei_tiles(a.tiles, a.size, start=0)
    mode = 1 if a.mode == 'wsi' else 0
    sf = float(opts.get('scale_factor', 2.0))
    print(f'checkpoint {a.checkpoint}: {sum(v.numel() for v in sd.values()) / 1e6:.2f} M parameters, {nc} classes; {len(tiles)} tiles of '
          f'{tiles.shape[1]}x{tiles.shape[2]}, scale_factor {sf}, channel mode {a.mode}')
    eng = Engine(sd, device=0, max_batch=min(a.batch, len(tiles)), tile=tiles.shape[1:3], num_classes=nc, **opts)
    t0 = time.time()
    got = eng(tiles, mode)
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    orc = O.Oracle(sd, num_classes=nc, score_thr=opts['score_thr'], max_per_img=opts['max_per_img'], scale=sf)
    t0 = time.time()
    ref, vals = [], []
    for i in range(0, len(tiles), 4):                     # oracle in small batches (memory)
        r, it = orc(tiles[i:i + 4], mode, keep=True)
        ref += r
        vals += P.oracle_paste_values(O, it, tiles.shape[1:3], sf)
    t_cpu = time.time() - t0
    bad = 0
    tot = dict(n_ref=0, n_got=0, matched=0, mask_px_flipped=0, masks_below_0999=0)
    for i, (r, g) in enumerate(zip(ref, got)):
        rep, fails = P.compare_strict(r, g, score_thr=opts['score_thr'], nms_iou=opts['nms_iou'], max_per_img=opts['max_per_img'], values=vals[i])
        for k in tot:
            tot[k] += rep[k]
        per_class = [len(b) for b in g[0]]
        print(f'tile {i:3d}: {P.fmt(rep)}; per class {per_class}')
        for e in rep['explained']:
            print('      tolerated:', e)
        for f in fails:
            print('      FAIL:', f)
        bad += bool(fails)
    print(f'total: oracle {tot["n_ref"]} instances, HIP {tot["n_got"]}, matched {tot["matched"]}; {tot["mask_px_flipped"]} mask pixels differ '
          f'({tot["masks_below_0999"]} masks below IoU 0.999); oracle {t_cpu:.1f} s on {torch.get_num_threads()} host threads, HIP {t_hip:.2f} s (first call, incl. upload)')
    print('PARITY OK' if not bad else f'PARITY FAILED on {bad} tile(s)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
