#!/usr/bin/env python
"""WSI tile inference with the reference's flags (tools/infer_wsi.py:309-356) over array inputs, sharded across GPUs.

    python tools/infer_wsi.py <source> <config> <checkpoint> [--patch_size 256 --step_size 192 --batch_size 16
                               --margin 2 --min_area 10 --save_dir out --gpus N]
    torchrun --nproc-per-node N tools/infer_wsi.py ...       (one rank per GPU; records gathered over RCCL)

<source>: .npy image (H,W,3) uint8 that is tiled on a grid, or .npz with `tiles` (N,P,P,3) and `coords` (N,2).
OpenSlide slides, tissue segmentation and the DSA/SQL writers of the reference are out of scope.
Output: <save_dir>/<name>.json — merged detections (box, score, label) after the cross-tile mask-IoU merge."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('source')
    p.add_argument('config')
    p.add_argument('checkpoint', nargs='?', default=None)
    p.add_argument('--device', default='cuda:0')
    p.add_argument('--patch_size', type=int, default=256)
    p.add_argument('--step_size', type=int, default=192)
    p.add_argument('--batch_size', type=int, default=16)
    p.add_argument('--num_workers', type=int, default=0)
    p.add_argument('--margin', type=int, default=2)
    p.add_argument('--min_area', type=int, default=10)
    p.add_argument('--mag', type=int, default=40)
    p.add_argument('--overlap_threshold', type=float, default=0.05)
    p.add_argument('--save_dir', default='wsi_out')
    return p.parse_args()


def main():
    args = parse_args()
    import torch
    from nuhtc_amd import parallel, wsi
    from nuhtc_amd.apis import init_detector
    rank, local_rank, world = parallel.init_from_env()
    if args.mag != 40:
        raise SystemExit('only --mag 40 (scale_factor 80/mag = 2.0) is supported by the engine')
    if args.source.endswith('.npz'):
        z = np.load(args.source)
        tiles, coords = z['tiles'], z['coords']
    else:
        tiles, coords = wsi.tile_grid(np.load(args.source), args.patch_size, args.step_size)
    lo, hi = parallel.shard_range(len(tiles), rank, world)
    model = init_detector(args.config, args.checkpoint, device=f'cuda:{local_rank}' if world > 1 else args.device, max_batch=args.batch_size)
    model.opts.update(margin=args.margin, min_area=args.min_area, mask_nms_thr=0.05)
    rec = wsi.infer_tiles(model, tiles[lo:hi], coords[lo:hi], args.batch_size)
    # one gather of fixed-width records; masks travel as (x0, y0, w, h) + bit-packed crop padded to 64x64 px
    n = len(rec['score'])
    F = 8 + 512
    buf = torch.zeros((n, F), dtype=torch.float32)
    for i in range(n):
        m, x0, y0 = rec['mask'][i]
        crop = np.zeros((64, 64), bool)
        crop[:min(64, m.shape[0]), :min(64, m.shape[1])] = m[:64, :64]
        buf[i, :4] = torch.from_numpy(rec['box'][i]).float()
        buf[i, 4], buf[i, 5], buf[i, 6], buf[i, 7] = rec['score'][i], rec['label'][i], x0, y0
        buf[i, 8:] = torch.from_numpy(np.packbits(crop).astype(np.float32))
    dev = torch.device('cuda', local_rank) if world > 1 and torch.cuda.is_available() else torch.device('cpu')
    parts = parallel.gather_records(buf.to(dev))
    if rank == 0:
        allrec = torch.cat([p.cpu() for p in parts]).numpy()
        full = dict(score=allrec[:, 4].tolist(), label=allrec[:, 5].astype(int).tolist(), box=[r[:4] for r in allrec], mask=[])
        for r in allrec:
            crop = np.unpackbits(r[8:].astype(np.uint8)).reshape(64, 64).astype(bool)
            ys, xs = np.nonzero(crop)
            crop = crop[:ys.max() + 1, :xs.max() + 1] if len(ys) else crop[:1, :1]
            full['mask'].append((crop, int(r[6]), int(r[7])))
        keep = wsi.merge_overlap(full, args.overlap_threshold)
        os.makedirs(args.save_dir, exist_ok=True)
        name = os.path.splitext(os.path.basename(args.source))[0]
        out = [dict(nuclei_id=int(k), bbox=[float(v) for v in allrec[i, :4]], score=float(allrec[i, 4]), label=int(allrec[i, 5]))
               for k, i in enumerate(keep)]
        with open(os.path.join(args.save_dir, name + '_merged.json'), 'w') as f:
            json.dump(out, f)
        print(f'{len(tiles)} tiles, {len(allrec)} detections after per-tile mask-NMS, {len(keep)} after cross-tile merge')


if __name__ == '__main__':
    main()
