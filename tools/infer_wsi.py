#!/usr/bin/env python
"""WSI tile inference with the reference's flags (tools/infer_wsi.py:309-356) over array inputs, sharded across GPUs.

    python tools/infer_wsi.py <source> <config> <checkpoint> [--patch_size 256 --step_size 192 --batch_size 16
                               --margin 2 --min_area 10 --save_dir out --mode qupath]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/infer_wsi.py ...   (one rank per GPU)

<source>: .npy image (H,W,3) uint8 (memory-mapped) that is tiled on a grid (np.arange(0, size, step), zero padded; with --seg only the
tissue found by nuhtc_amd.tissue is tiled, as the reference's seg_and_patch does; with --coords the given level-0 origins, the role
of the reference's patches/<name>.h5), a store directory (nuhtc_amd.tilestore.write_store), or .npz with `tiles` (N,P,P,3) and
`coords` (N,2).  Every rank cuts only the tiles of its own shard.  OpenSlide / HDF5 reading is out of scope (neither library
exists offline).
Output (like the reference, :659-693), for every detection that survives the per-tile filter + mask-NMS:
  --mode qupath : <save_dir>/nuclei/<name>/<name>.geojson and <name>_point.geojson (flat lists of QuPath features); run
                  tools/nuclei_merge.py on the .geojson for the cross-tile merge (or pass --merge to do it here on rank 0)
  --mode dsa    : <name>_dsa.json (HistomicsUI polyline elements)
  --mode coco   : coco_nuclei.json (per-tile images + RLE annotations) and <save_dir>/imgs/<name>/<annidx>.png
  --mode sql    : <name>_dql.db (contour table + R-tree)
  --mode all    : everything."""
import argparse
import json
import os
import sys

# 16 hardware queues instead of the HIP runtime's 4 (read when the runtime initialises; an exported value wins): see tools/bench_wsi.py
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

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
    p.add_argument('--mode', default='qupath', choices=['qupath', 'dsa', 'coco', 'sql', 'all'])
    p.add_argument('--seg', action='store_true', help='segment tissue first and tile only the tissue contours (reference --seg --patch)')
    p.add_argument('--seg_downsample', type=int, default=64, help='downsample factor of the segmentation level (reference: pyramid level nearest 64x)')
    p.add_argument('--coords', default=None, help='coordinate file of a .npy slide: .npy (N,2) or .npz with `coords` [+ `patch_size`] -- the role of the reference\'s patches/<name>.h5')
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
    # tile source with per-rank lazy reads (nuhtc_amd.tilestore): the slide stays memory-mapped, every rank knows all
    # coordinates (16 bytes per tile) and cuts only the tiles of its own shard; tissue segmentation runs on rank 0 only
    from nuhtc_amd import tilestore
    coords_fn = None
    if args.seg:
        def coords_fn(slide):
            c = [None]
            if rank == 0:
                from nuhtc_amd import tissue
                c[0], conts, _ = tissue.tissue_tile_coords(slide, args.patch_size, args.step_size, scale=args.seg_downsample)
                print(f'tissue segmentation: {len(conts)} contour(s), {len(c[0])} tiles')
            if world > 1:
                import torch.distributed as dist
                dist.broadcast_object_list(c, src=0)
            return c[0]
    bag = tilestore.open_source(args.source, args.patch_size, args.step_size, coords=args.coords, coords_fn=coords_fn)
    coords = bag.coords
    lo, hi = parallel.shard_range(len(bag), rank, world)
    tiles = bag.read(lo, hi)                                  # this rank's tiles only
    from nuhtc_amd.config import Config, set_test_scale_factor
    cfg = Config.fromfile(args.config)
    sf = set_test_scale_factor(cfg, args.mag)          # reference :416-419: scale_factor = 80 / mag
    if rank == 0:
        print('scale_factor: ', sf)
    model = init_detector(cfg, args.checkpoint, device=f'cuda:{local_rank}' if world > 1 else args.device, max_batch=args.batch_size)
    model.CLASSES = ('T', 'I', 'C', 'D', 'E')[:model.opts['num_classes']]
    model.opts.update(margin=args.margin, min_area=args.min_area, mask_nms_thr=0.05)
    rec = wsi.infer_tiles(model, tiles, coords[lo:hi], args.batch_size)
    # contours are traced on the rank that owns the tile; two variable-length gathers: records, then ring vertices
    rings = rec['ring']                                              # traced on the GPU (nuhtc_mask_contours)
    keep = [i for i, r in enumerate(rings) if len(r) >= 3]          # reference :536 tests the CLOSED contour (mask2inst appends the first point): only one-pixel contours go
    n = len(keep)
    want = lambda m: args.mode in (m, 'all')
    P = bag.patch_size
    rles = []
    if want('coco'):                                                  # RLE of the instance inside its tile (:611-613)
        from nuhtc_amd import cocomask
        for i in keep:
            crop, x0, y0 = rec['mask'][i]
            ox, oy = (int(v) for v in coords[lo + rec['tile'][i]])
            full = np.zeros((P, P), np.uint8)
            full[y0 - oy:y0 - oy + crop.shape[0], x0 - ox:x0 - ox + crop.shape[1]] = crop
            rles.append(cocomask.encode(full)['counts'].encode('ascii'))
    # the one exchange of the path: every rank's records (head, ring vertices, mask crops, RLE strings) in a single all-gather
    dev = torch.device('cuda', local_rank) if world > 1 and torch.cuda.is_available() else torch.device('cpu')
    gathered = parallel.gather_blobs([t.to(dev) for t in wsi.pack_records(rec, keep, tile_base=lo, rles=rles)])
    heads = [g[0] for g in gathered]
    vparts = [g[1] for g in gathered]
    bparts = [g[4] if want('coco') else None for g in gathered]
    if rank == 0:
        from nuhtc_amd import outputs
        name = os.path.splitext(os.path.basename(os.path.normpath(args.source)))[0]
        out_dir = os.path.join(args.save_dir, 'nuclei', name)
        os.makedirs(out_dir, exist_ok=True)
        feats, points, dsa, annts, per_tile = [], [], [], [], {}
        sql = outputs.SqlContourWriter(os.path.join(out_dir, name + '_dql.db')) if want('sql') else None
        for h, v, bl in zip(heads, vparts, bparts):
            h, v = h.cpu().numpy(), v.cpu().numpy()
            bl = bl.cpu().numpy().tobytes() if bl is not None else b''
            off = boff = 0
            for row in h:
                nv, annidx, nb = int(row[6]), int(row[7]), int(row[8])
                ring = v[off:off + nv].astype(np.int64)
                off += nv
                label, score = int(row[5]), float(row[4])
                elementidx = len(per_tile.setdefault(annidx, []))
                per_tile[annidx].append(label)
                if want('qupath'):
                    feats.append(contours.feature(ring, label, score, model.CLASSES))
                    points.append(contours.point_feature(row[:4], label, score, model.CLASSES))
                if want('dsa'):
                    dsa.append(outputs.dsa_element(ring, label, model.CLASSES))
                if want('coco'):
                    rle = {'size': [P, P], 'counts': bl[boff:boff + nb].decode('ascii')}
                    boff += nb
                    bbox = cocomask.to_bbox(rle)
                    annts.append({'bbox': bbox, 'area': bbox[2] * bbox[3], 'image_id': annidx, 'category_id': label, 'id': len(annts),
                                  'iscrowd': 0, 'segmentation': rle})
                if sql:
                    sql.add(annidx, elementidx, ring, label, score, model.CLASSES)
        msg = f'{len(bag)} tiles on {world} rank(s): {sum(len(v) for v in per_tile.values())} nuclei after per-tile mask-NMS'
        if want('qupath'):
            outputs.write_json(os.path.join(out_dir, name + '.geojson'), feats)
            outputs.write_json(os.path.join(out_dir, name + '_point.geojson'), points)
            if args.merge:
                kept = wsi.merge_gathered(gathered, args.overlap_threshold, device=local_rank if world > 1 else (torch.device(args.device).index or 0))
                merged = [feats[i] for i in kept]
                outputs.write_json(os.path.join(out_dir, name + '_merged.geojson'), merged)
                msg += f', {len(merged)} after the cross-tile merge'
        if want('dsa'):
            outputs.write_json(os.path.join(out_dir, name + '_dsa.json'), outputs.dsa_document(dsa))
        if want('coco'):
            from PIL import Image
            img_dir = os.path.join(args.save_dir, 'imgs', name)
            os.makedirs(img_dir, exist_ok=True)
            imgs = []
            for annidx in sorted(per_tile):
                imgs.append(outputs.coco_tile_image(annidx, P, P, per_tile[annidx], model.CLASSES))
                Image.fromarray(bag[annidx][0]).save(os.path.join(img_dir, f'{annidx}.png'))
            outputs.write_json(os.path.join(out_dir, 'coco_nuclei.json'),
                               {'images': imgs, 'annotations': annts, 'categories': outputs.coco_categories(model.CLASSES)})
        if sql:
            sql.close()
        print(msg)


if __name__ == '__main__':
    main()
