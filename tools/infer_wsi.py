#!/usr/bin/env python
"""WSI tile inference with the reference's flags (tools/infer_wsi.py:309-356) over array inputs, sharded across GPUs.

    python tools/infer_wsi.py <source> <config> <checkpoint> [--patch_size 256 --step_size 192 --batch_size 16
                               --margin 2 --min_area 10 --save_dir out --mode qupath]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/infer_wsi.py ...   (one rank per GPU)

<source>: .npy image (H,W,3) uint8 that is tiled on a grid (np.arange(0, size, step), zero padded), or .npz with
`tiles` (N,P,P,3) and `coords` (N,2).  OpenSlide reading, tissue segmentation and the DSA/SQL/COCO writers of the
reference are out of scope.
Output (like the reference, :659-664): <save_dir>/nuclei/<name>/<name>.geojson and <name>_point.geojson — flat lists of
QuPath features of every detection that survives the per-tile filter + mask-NMS; run tools/nuclei_merge.py on the
.geojson for the cross-tile merge (or pass --merge to do it here on rank 0)."""
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
    p.add_argument('--mode', default='qupath', choices=['qupath'])
    p.add_argument('--merge', action='store_true', help='also run the cross-tile merge (nuclei_merge.py) on rank 0')
    p.add_argument('--overlap_threshold', type=float, default=0.05)
    p.add_argument('--save_dir', default='wsi_out')
    return p.parse_args()


def main():
    args = parse_args()
    import torch
    from nuhtc_amd import contours, parallel, wsi
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
    model.CLASSES = ('T', 'I', 'C', 'D', 'E')[:model.opts['num_classes']]
    model.opts.update(margin=args.margin, min_area=args.min_area, mask_nms_thr=0.05)
    rec = wsi.infer_tiles(model, tiles[lo:hi], coords[lo:hi], args.batch_size)
    # contours are traced on the rank that owns the tile; two variable-length gathers: records, then ring vertices
    rings = [contours.mask_to_ring(m, origin=(x0, y0)) for (m, x0, y0) in rec['mask']]
    keep = [i for i, r in enumerate(rings) if len(r) >= 4]          # reference drops contours with < 3 points (:536)
    n = len(keep)
    head = torch.zeros((n, 7), dtype=torch.float64)
    for k, i in enumerate(keep):
        head[k, :4] = torch.from_numpy(rec['box'][i])
        head[k, 4], head[k, 5], head[k, 6] = rec['score'][i], rec['label'][i], len(rings[i])
    verts = torch.from_numpy(np.concatenate([rings[i] for i in keep], 0).astype(np.float64)) if n else torch.zeros((0, 2), dtype=torch.float64)
    dev = torch.device('cuda', local_rank) if world > 1 and torch.cuda.is_available() else torch.device('cpu')
    heads = parallel.gather_records(head.to(dev))
    vparts = parallel.gather_records(verts.to(dev))
    if rank == 0:
        feats, points = [], []
        for h, v in zip(heads, vparts):
            h, v = h.cpu().numpy(), v.cpu().numpy()
            off = 0
            for row in h:
                nv = int(row[6])
                ring = v[off:off + nv].astype(np.int64)
                off += nv
                feats.append(contours.feature(ring, int(row[5]), float(row[4]), model.CLASSES))
                points.append(contours.point_feature(row[:4], int(row[5]), float(row[4]), model.CLASSES))
        name = os.path.splitext(os.path.basename(args.source))[0]
        out_dir = os.path.join(args.save_dir, 'nuclei', name)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, name + '.geojson'), 'w') as f:
            json.dump(feats, f)
        with open(os.path.join(out_dir, name + '_point.geojson'), 'w') as f:
            json.dump(points, f)
        msg = f'{len(tiles)} tiles on {world} rank(s): {len(feats)} nuclei after per-tile mask-NMS'
        if args.merge:
            merged = contours.merge_features(feats, args.overlap_threshold, 'probability')
            with open(os.path.join(out_dir, name + '_merged.geojson'), 'w') as f:
                json.dump(merged, f)
            msg += f', {len(merged)} after the cross-tile merge'
        print(msg)


if __name__ == '__main__':
    main()
