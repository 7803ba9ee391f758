#!/usr/bin/env python
"""WSI inference with the reference's command line (tools/infer_wsi.py:309-356: same flags, same defaults) over array slides,
sharded across GPUs.

    python tools/infer_wsi.py <source> <config> <checkpoint> --patch --seg --stitch --patch_size 256 --step_size 192 \
                               --batch_size 16 --save_dir out --mode qupath [--slide_ext .npy]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/infer_wsi.py ...   (one rank per GPU)

<source> is the reference's folder of slides: every file in it goes through `seg_and_patch` (nuhtc_amd/slides.py: process list
-> <save_dir>/process_list_autogen.csv, `--seg` tissue segmentation, masks/<id>.png, `--patch` tile coordinates ->
patches/<id>.npz, `--stitch` stitches/<id>.jpg; a slide whose coordinate file -- that .npz or the reference's patches/<id>.h5 -- exists is skipped unless --no_auto_skip), then every
slide of the process list that has a coordinate file and no <id>_merged.geojson yet (:445-458) is tiled, inferred and written.
A slide is a level-0 RGB array (`.npy`, memory-mapped; pass `--slide_ext .npy`) or a store directory: OpenSlide / HDF5 do not
exist offline.  Beyond the reference, <source> may be ONE slide: a `.npy` file (`--patch` is then implied: the
grid np.arange(0, size, step), or the tissue tiles with --seg, or the origins of --coords), a store directory
(nuhtc_amd.tilestore.write_store) or an `.npz` with `tiles` (N,P,P,3) + `coords` (N,2).  Every rank cuts only the tiles of its
own shard; tissue segmentation runs on rank 0.
Output (like the reference, :659-693), for every detection that survives the per-tile filter + mask-NMS:
  --mode qupath : <save_dir>/nuclei/<id>/<id>.geojson and <id>_point.geojson (flat lists of QuPath features); run
                  tools/nuclei_merge.py on the .geojson for the cross-tile merge (or pass --merge to do it here on rank 0)
  --mode dsa    : <id>_dsa.json (HistomicsUI polyline elements)
  --mode coco   : coco_nuclei.json (per-tile images + RLE annotations) and <save_dir>/imgs/<id>/<annidx>.png
  --mode sql    : <id>_dql.db (contour table + R-tree)
  --mode all    : everything.
  --det         : <save_dir>/<id>/infer/img_<x>_<y>.jpg overlays of every tile with detections (score >= --score-thr; :504-512)."""
import argparse
import os
import sys

# 16 hardware queues instead of the HIP runtime's 4 (read when the runtime initialises; an exported value wins): see tools/bench_wsi.py
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    """The reference's parser (tools/infer_wsi.py:309-356), flag for flag and default for default (pinned by
    tests/test_cli_parity.py against a table read off the reference), plus this build's own flags at the end.
    allow_abbrev=False: with abbreviations `--patch` would be taken for a prefix of `--patch_size`."""
    p = argparse.ArgumentParser(allow_abbrev=False)
    p.add_argument('source', help='path to folder containing raw wsi image files (or one array slide / tile store)')
    p.add_argument('config', help='Config file')
    p.add_argument('checkpoint', help='Checkpoint file')
    p.add_argument('--device', default='cuda:0', help='Device used for inference')
    p.add_argument('--score-thr', type=float, default=0.35, help='score threshold')
    p.add_argument('--async-test', action='store_true', help='whether to set async options for async inference.')
    p.add_argument('--step_size', type=int, default=256, help='step_size')
    p.add_argument('--patch_size', type=int, default=256, help='patch_size')
    p.add_argument('--patch', default=False, action='store_true')
    p.add_argument('--seg', default=False, action='store_true')
    p.add_argument('--stitch', default=False, action='store_true')
    p.add_argument('--no_auto_skip', default=False, action='store_true')
    p.add_argument('--save_dir', type=str, help='directory to save processed data')
    p.add_argument('--preset', default=None, type=str, help='predefined profile of default segmentation and filter parameters (.csv)')
    p.add_argument('--patch_level', type=int, default=0, help='downsample level at which to patch')
    p.add_argument('--mag', '--magnification', type=int, default=40, help='magnification for the slide', dest='mag')
    p.add_argument('--batch_size', type=int, default=32, help='batch size of image dataset during inference')
    p.add_argument('--num_workers', type=int, default=8, help='number workers of image dataset during inference')
    p.add_argument('--margin', type=int, default=0, help='discard the contour which distance is less than margin number pixels to edges')
    p.add_argument('--min_area', type=int, default=10, help='discard the area less than min_area')
    p.add_argument('--process_list', type=str, default=None, help='name of list of images to process with parameters (.csv)')
    p.add_argument('--slide_ext', type=str, default='.svs', help='ext name of wsi')
    p.add_argument('--mode', type=str, default='qupath', help='mode of save format')
    p.add_argument('--det', default=False, action='store_true')
    # ---- not in the reference
    p.add_argument('--seg_downsample', type=int, default=64, help='downsample of the (virtual) pyramid level seg_level / vis_level -1 resolve to (reference: the level nearest 64x)')
    p.add_argument('--coords', default=None, help="coordinate file of a single .npy slide: .npy (N,2), .npz with `coords` [+ `patch_size`], or the reference's own patches/<id>.h5")
    p.add_argument('--gpus', type=int, default=1, help='one rank per GPU: the tool starts itself N times under torch.distributed.run (a child process) unless a launcher already did')
    p.add_argument('--merge', action='store_true', help='also run the cross-tile merge (tools/nuclei_merge.py) on rank 0 -> <id>_merged.geojson')
    p.add_argument('--overlap_threshold', type=float, default=0.05)
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def run_slide(args, model, bag, slide_id, rank, local_rank, world):
    """The reference's per-slide loop (:460-693): tiles -> detections -> per-tile filter + mask-NMS -> records -> the one gather ->
    rank 0 writes the documents.  The tile list is sharded in contiguous blocks over the ranks."""
    import torch
    from nuhtc_amd import contours, parallel, wsi
    coords = bag.coords
    lo, hi = parallel.shard_range(len(bag), rank, world)
    tiles = bag.view(lo, hi)                                  # this rank's tiles only, cut / decoded a batch at a time while earlier batches run
    rec = wsi.infer_tiles(model, tiles, coords[lo:hi], args.batch_size)
    # contours are traced on the rank that owns the tile; two variable-length gathers: records, then ring vertices
    rings = rec['ring']                                              # traced on the GPU (nuhtc_mask_contours)
    keep = [i for i, r in enumerate(rings) if len(r) >= 3]          # reference :536 tests the CLOSED contour (mask2inst appends the first point): only one-pixel contours go
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
    if args.det:                                                      # :504-512, per tile with detections; a second pass through the API (debug output)
        from nuhtc_amd.apis import inference_detector, save_result
        det_dir = os.path.join(args.save_dir, slide_id, 'infer')
        os.makedirs(det_dir, exist_ok=True)
        for i0 in range(0, len(tiles), args.batch_size):
            batch = [t for t in tiles[i0:i0 + args.batch_size]]
            for k, res in enumerate(inference_detector(model, batch)):
                if sum(len(b) for b in res[0]):
                    x, y = (int(v) for v in coords[lo + i0 + k])
                    save_result(model, batch[k], res, score_thr=args.score_thr, out_file=os.path.join(det_dir, f'img_{x}_{y}.jpg'))
    # the one exchange of the path: every rank's records (head, ring vertices, mask crops, RLE strings) in a single all-gather
    dev = torch.device('cuda', local_rank) if world > 1 and torch.cuda.is_available() else torch.device('cpu')
    parts = wsi.pack_records(rec, keep, tile_base=lo, rles=rles)
    if want('qupath'):
        # every rank writes the GeoJSON text of ITS records (the reference's one Python loop over all nuclei, :533-585 + json.dump :659-664,
        # is seconds per slide on the writing rank); the bytes travel in the same gather and rank 0 only concatenates
        h0, v0 = parts[0].numpy(), parts[1].numpy()
        lab = h0[:, 5].astype(np.int32)
        ptxt, pstart = contours.ring_features_text(v0, h0[:, 6].astype(np.int64), lab, h0[:, 4], model.CLASSES)
        parts += [torch.from_numpy(ptxt), torch.from_numpy(pstart), torch.from_numpy(contours.point_features_text(h0[:, :4], lab, h0[:, 4], model.CLASSES))]
    parts.append(torch.tensor([rank], dtype=torch.int32))           # who sent it (printed by rank 0)
    gathered = parallel.gather_blobs([t.to(dev) for t in parts])
    heads = [g[0] for g in gathered]
    vparts = [g[1] for g in gathered]
    bparts = [g[4] if want('coco') else None for g in gathered]
    if rank != 0:
        return
    from nuhtc_amd import outputs
    name = slide_id
    out_dir = os.path.join(args.save_dir, 'nuclei', name)
    os.makedirs(out_dir, exist_ok=True)
    dsa, annts, per_tile = [], [], {}
    sql = outputs.SqlContourWriter(os.path.join(out_dir, name + '_dql.db')) if want('sql') else None
    n_records = int(sum(len(h) for h in heads))
    for h, v, bl in zip(heads, vparts, bparts):
        if not (want('dsa') or want('coco') or sql):       # the per-record loop serves the other document kinds only
            break
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
    msg = f'{len(bag)} tiles on {world} rank(s): {n_records} nuclei after per-tile mask-NMS'
    if world > 1:
        msg += f' (records of ranks {sorted(int(g[-1][0]) for g in gathered)}, tiles per rank {[parallel.shard_range(len(bag), r, world)[1] - parallel.shard_range(len(bag), r, world)[0] for r in range(world)]})'
    if want('qupath'):
        body, start = contours.concat_feature_texts([g[5].cpu().numpy() for g in gathered], [g[6].cpu().numpy() for g in gathered])
        outputs.write_text_list(os.path.join(out_dir, name + '.geojson'), body)
        outputs.write_text_list(os.path.join(out_dir, name + '_point.geojson'), contours.concat_feature_texts([g[7].cpu().numpy() for g in gathered])[0])
        if args.merge:
            kept = wsi.merge_gathered(gathered, args.overlap_threshold, device=local_rank if world > 1 else (torch.device(args.device).index or 0))
            outputs.write_text_list(os.path.join(out_dir, name + '_merged.geojson'), contours.join_features_text(body, start, kept))
            msg += f', {len(kept)} after the cross-tile merge'
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


def main(argv=None):
    args = parse_args(argv)
    if args.save_dir is None:
        raise SystemExit('--save_dir is required (the reference joins it with "patches" / "masks" / "stitches" at once, :360-362)')
    if args.mode not in ('qupath', 'dsa', 'coco', 'sql', 'all'):
        raise SystemExit(f'--mode {args.mode}: one of qupath, dsa, coco, sql, all')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # no launcher: become one (before anything here touches the GPU)
        from nuhtc_amd import parallel
        raise SystemExit(parallel.self_launch(args.gpus, __file__, sys.argv[1:] if argv is None else argv))
    import torch
    from nuhtc_amd import parallel, slides, tilestore
    from nuhtc_amd.apis import init_detector
    # the process group is formed AFTER seg_and_patch: rank 0's host phase (segmentation, masks, patching, stitching of every slide of the
    # folder) has no time bound, and a rank waiting in an RCCL barrier is aborted by the watchdog after 10 minutes
    rank, local_rank, world = parallel.env_ranks()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    say = print if rank == 0 else (lambda *a, **k: None)
    if args.async_test:
        say('--async-test: accepted; the reference never reads it in this tool either')
    say(f'--num_workers {args.num_workers}: accepted; tiles are cut from memory-mapped arrays by the rank that owns them (no DataLoader workers)')
    patch_save_dir = os.path.join(args.save_dir, 'patches')
    mask_save_dir = os.path.join(args.save_dir, 'masks')
    stitch_save_dir = os.path.join(args.save_dir, 'stitches')
    process_list = os.path.join(args.save_dir, args.process_list) if args.process_list else None
    directories = {'source': args.source, 'save_dir': args.save_dir, 'patch_save_dir': patch_save_dir, 'mask_save_dir': mask_save_dir,
                   'stitch_save_dir': stitch_save_dir}
    src = os.path.normpath(args.source)
    is_store = os.path.isdir(src) and os.path.isfile(os.path.join(src, 'slide.npy')) and os.path.isfile(os.path.join(src, 'coords.npy'))
    single = is_store or os.path.isfile(src)                  # beyond the reference: ONE slide instead of a folder
    if rank == 0:
        for key, val in directories.items():
            print('{} : {}'.format(key, val))
            if key not in ['source']:
                os.makedirs(val, exist_ok=True)
    seg_params, filter_params, vis_params, patch_params = slides.default_parameters(args.preset)
    say({'seg_params': seg_params, 'filter_params': filter_params, 'patch_params': patch_params, 'vis_params': vis_params})
    from nuhtc_amd.config import Config, set_test_scale_factor
    cfg = Config.fromfile(args.config)
    sf = set_test_scale_factor(cfg, args.mag)          # reference :416-419: scale_factor = 80 / mag
    say('scale_factor: ', sf)
    # this tool owns its process: the submitting thread goes onto the GPU's NUMA node (NUHTC_HOST_AFFINITY=0 leaves it alone)
    model = init_detector(cfg, args.checkpoint, device=f'cuda:{local_rank}' if world > 1 else args.device, max_batch=args.batch_size,
                          bind_host=os.environ.get('NUHTC_HOST_AFFINITY', '1') != '0')
    model.CLASSES = ('T', 'I', 'C', 'D', 'E')[:model.opts['num_classes']]
    model.opts.update(margin=args.margin, min_area=args.min_area, mask_nms_thr=0.05)

    # ---- seg_and_patch on rank 0 (:430-435), everybody else waits for the process list and the coordinate files
    jobs = []                                                 # (slide_id, bag factory)
    if single and (is_store or src.endswith('.npz')):         # the tiles / coordinates come with the source: nothing to segment or patch
        slide_id = os.path.splitext(os.path.basename(src))[0]
        jobs.append((slide_id, lambda: tilestore.open_source(src, args.patch_size, args.step_size)))
    else:
        folder, names, ext = (os.path.dirname(src) or '.', [os.path.basename(src)], os.path.splitext(src)[1]) if single else (src, None, args.slide_ext)
        if rank == 0:
            if single and args.coords is not None:            # explicit origins: they ARE the coordinate file
                c, ps = tilestore._load_coords(args.coords)
                sid = os.path.splitext(names[0])[0]
                slides.save_coords(slides.coords_path(patch_save_dir, sid), c, ps or args.patch_size, 0, sid)
            slides.seg_and_patch(folder, args.save_dir, patch_save_dir, mask_save_dir, stitch_save_dir, seg_params=seg_params,
                                 filter_params=filter_params, vis_params=vis_params, patch_params=patch_params,
                                 patch_size=args.patch_size, step_size=args.step_size, seg=args.seg, use_default_params=False,
                                 save_mask=True, stitch=args.stitch, patch_level=args.patch_level, patch=args.patch or single,
                                 process_list=process_list, no_auto_skip=args.no_auto_skip or (single and args.coords is None),
                                 slides=names, seg_downsample=args.seg_downsample)
        parallel.host_phase_done(args.save_dir, rank, world)      # the other ranks poll a marker file (no collective, no timeout)
        for entry in slides.slide_list(args.save_dir):        # Dataset_All_Bags over process_list_autogen.csv (:437-439)
            slide_id = entry.split(ext)[0] if ext else entry
            if not slides.has_coords(patch_save_dir, slide_id):     # patches/<id>.npz, or the reference's own patches/<id>.h5
                say(f'\nskip {slide_id} due to no coord file')
                continue
            spath = os.path.join(folder, entry)                # the list holds file names; the reference rebuilds them as slide_id + slide_ext (:452)
            if not os.path.exists(spath):
                spath = os.path.join(folder, slide_id + ext)

            def make(spath=spath, slide_id=slide_id):
                c, ps, lvl = slides.load_coords(patch_save_dir, slide_id)
                if lvl != 0:                                   # (a reference-made .h5 can carry another level; tiles are cut from level 0 here)
                    raise SystemExit(f'{slide_id}: coordinate file was made for patch_level {lvl}; only patch_level 0 is supported')
                return tilestore.TileBag(slides.open_array_slide(spath), c, ps)
            jobs.append((slide_id, make))
    parallel.init_from_env()
    total = len(jobs)
    for k, (slide_id, make) in enumerate(jobs):
        say('\nprogress: {}/{}'.format(k, total))
        say(slide_id)
        # a slide that already has its merged file is done (:456-458); in single-slide mode the run is the request: always redone
        if not single and os.path.exists(os.path.join(args.save_dir, 'nuclei', slide_id, f'{slide_id}_merged.geojson')):
            say(f'skip {slide_id} due to existing results')
            continue
        run_slide(args, model, make(), slide_id, rank, local_rank, world)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                                        # every rank is past the marker
        parallel.host_phase_cleanup(args.save_dir, rank, world)


if __name__ == '__main__':
    main()
