"""BASELINE configs[2] end to end: a synthetic slide (G x G tiles of 256x256 at stride 192 over one S-nuclei canvas),
the tile stream sharded in contiguous blocks across one process per GPU, one variable-length gather of the detection
records and the cross-tile mask merge on rank 0 (tools/infer_wsi.py:460-531 + tools/nuclei_merge.py:62-174).

    python tools/bench_wsi.py --grid 100                      # 10 000 tiles, one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_wsi.py --grid 100

Prints one JSON line on rank 0.  Inputs start in host memory (pageable tiles, as the slide reader delivers them), so
the inference rate here includes H2D copies, device contour tracing, D2H of the kept masks and the host-side record
building -- it is the slide-level rate, not bench.py's HBM-resident `value`."""
import argparse
import json
import os
import sys
import time

# 16 hardware queues instead of the HIP runtime's 4 (read when the runtime initialises; an exported value wins): the slide loop keeps
# four one-stream engines busy (nuhtc_amd.pipeline) and also uses the caller's stream -- with four queues that stream shares a
# queue with an engine: 1.57k tiles/s slide-level against 1.80k with 16 queues (three engines on four queues: 1.73k)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=100, help='tiles per side (default 100 -> 10 000 tiles)')
    ap.add_argument('--batch_size', type=int, default=16)
    ap.add_argument('--depth', type=int, default=4, help='engines (slots) per GPU; each holds up to two batches')
    ap.add_argument('--overlap_threshold', type=float, default=0.05)
    ap.add_argument('--workers', type=int, default=16, help='host processes rendering the synthetic canvas')
    ap.add_argument('--config', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'))
    ap.add_argument('--checkpoint', default=None, help='pannuke.pth when available; seeded synthetic weights otherwise')
    ap.add_argument('--svs', default=None, choices=['jpeg', 'lzw'], help='write the canvas as an Aperio-layout .svs first (240-pixel tiles, this compression; not timed) '
                    'and feed the loop from the FILE through libtiff (nuhtc_amd.tiffslide) instead of from the array in memory')
    args = ap.parse_args()
    from nuhtc_amd import parallel, synth
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    G = args.grid
    lo, hi = parallel.shard_range(G * G, rank, world)
    t0 = time.perf_counter()
    t_svs, svs = None, None
    if args.svs:
        # the slide is a FILE every rank opens (as tools/infer_wsi.py's ranks do): rank 0 renders the whole canvas and writes it (not timed),
        # the others wait for it -- a marker file, before the process group exists (nuhtc_amd.parallel.host_phase_done) -- and each rank
        # decodes only the tiles of its own shard
        from nuhtc_amd import tiffslide, tilestore
        svs = f'/tmp/nuhtc_bench_wsi_{parallel.job_token()}.svs'
        if rank == 0:
            band, y0 = synth.nuclei_canvas_parallel(G, rows=(0, G), workers=args.workers)
            t_canvas = time.perf_counter() - t0
            t0 = time.perf_counter()
            tiffslide.write_pyramid(svs, band, levels=1, tile=240, compression=args.svs, quality=70, thumbnail=False,
                                    description=f'Aperio Image Library (synthetic)\n{band.shape[1]}x{band.shape[0]} (240x240) JPEG/RGB Q=70|AppMag = 40|MPP = 0.2500')
            t_svs = time.perf_counter() - t0
            del band
        else:
            t_canvas = 0.0
        parallel.host_phase_done('/tmp', rank, world)
        coords_all = synth.CanvasTiles.grid_coords(G)
        tiles = tilestore.TileBag(tiffslide.TiffSlide(svs), coords_all, 256).view(lo, hi)
        tiles.coords = coords_all[lo:hi]
    else:
        band, y0 = synth.nuclei_canvas_parallel(G, rows=(lo // G, (hi - 1) // G + 1), workers=args.workers)   # before any GPU call: the pool forks
        t_canvas = time.perf_counter() - t0
        tiles = synth.CanvasTiles(band, y0, G, lo, hi)

    import torch
    from nuhtc_amd import weights, wsi
    from nuhtc_amd.apis import init_detector
    rank, local_rank, world = parallel.init_from_env()
    ck = args.checkpoint
    if ck is None:
        ck = f'/tmp/nuhtc_bench_wsi_{os.getpid()}.pth'
        torch.save(dict(state_dict=weights.bench_state_dict(0)), ck)
    model = init_detector(args.config, ck, device=f'cuda:{local_rank}', max_batch=args.batch_size, bind_host=os.environ.get('NUHTC_HOST_AFFINITY', '1') != '0')   # this script owns its process
    model.opts.update(margin=2, min_area=10, mask_nms_thr=0.05)
    dev = torch.device('cuda', local_rank)
    wsi.infer_tiles(model, tiles[0:args.depth * args.batch_size], tiles.coords[:args.depth * args.batch_size], args.batch_size, args.depth)   # warm-up
    sync = lambda: (torch.distributed.barrier() if world > 1 else None, torch.cuda.synchronize(dev))
    sync()
    t0 = time.perf_counter()
    rec = wsi.infer_tiles(model, tiles, tiles.coords, args.batch_size, args.depth)
    sync()
    t_infer = time.perf_counter() - t0

    # one gather of the records (boxes, scores, rings, bit-packed mask crops) AND of the GeoJSON text every rank wrote for its own records
    # (tools/infer_wsi.py run_slide), merge on rank 0's GPU (polygon IoU), the three QuPath documents written by rank 0
    from nuhtc_amd import contours, outputs
    classes = ('T', 'I', 'C', 'D', 'E')
    t0 = time.perf_counter()
    parts = wsi.pack_records(rec)
    t1 = time.perf_counter()
    h0, v0 = parts[0].numpy(), parts[1].numpy()
    lab = h0[:, 5].astype(np.int32)
    ptxt, pstart = contours.ring_features_text(v0, h0[:, 6].astype(np.int64), lab, h0[:, 4], classes)
    parts += [torch.from_numpy(ptxt), torch.from_numpy(pstart), torch.from_numpy(contours.point_features_text(h0[:, :4], lab, h0[:, 4], classes))]
    t_text = time.perf_counter() - t1
    gathered = parallel.gather_blobs([t.to(dev) for t in parts])       # ONE all-gather of every rank's records and text
    sync()
    t_gather = time.perf_counter() - t0
    if rank == 0:
        t0 = time.perf_counter()
        kept = wsi.merge_gathered(gathered, args.overlap_threshold, device=local_rank)
        allp = np.concatenate([g[2].cpu().numpy() for g in gathered], 0)
        t_merge = time.perf_counter() - t0
        t0 = time.perf_counter()
        doc_dir = f'/tmp/nuhtc_bench_wsi_docs_{os.getpid()}'
        os.makedirs(doc_dir, exist_ok=True)
        body, start = contours.concat_feature_texts([g[5].cpu().numpy() for g in gathered], [g[6].cpu().numpy() for g in gathered])
        docs = {'slide.geojson': body, 'slide_point.geojson': contours.concat_feature_texts([g[7].cpu().numpy() for g in gathered])[0],
                'slide_merged.geojson': contours.join_features_text(body, start, kept)}
        for fname, text in docs.items():
            outputs.write_text_list(os.path.join(doc_dir, fname), text)
        t_docs = time.perf_counter() - t0
        doc_bytes = {k: int(len(v)) + 2 for k, v in docs.items()}
        for fname in docs:
            os.remove(os.path.join(doc_dir, fname))
        os.rmdir(doc_dir)
        total = G * G
        print(json.dumps({
            'workload': f'synthetic WSI, {G}x{G} tiles of 256x256 at stride 192 (BASELINE configs[2]), batch {args.batch_size}, {args.depth} batches in flight per GPU',
            'tiles': total, 'n_gpus': world, 'infer_s': round(t_infer, 3), 'tiles_per_s_inference': round(total / t_infer, 1),
            'gather_pack_s': round(t_gather, 3), 'geojson_text_s_this_rank': round(t_text, 3), 'merge_s': round(t_merge, 4),
            'tiles_per_s_end_to_end': round(total / (t_infer + t_gather + t_merge), 1),
            'documents_s': round(t_docs, 3), 'document_bytes': doc_bytes,
            'tiles_per_s_to_documents': round(total / (t_infer + t_gather + t_merge + t_docs), 1),
            'documents_note': 'slide.geojson, slide_point.geojson, slide_merged.geojson as tools/infer_wsi.py --mode qupath --merge writes them: every rank serialises its own '
                              'records (inside gather_pack_s), rank 0 concatenates, selects the merged records and writes the files (documents_s)',
            'detections_after_tile_nms': int(len(allp)), 'detections_after_merge': int(len(kept)),
            'canvas_render_s_host': round(t_canvas, 1), 'weights': 'pannuke.pth' if args.checkpoint else 'seeded synthetic',
            **({'tile_source': f'{args.svs}-compressed .svs of {os.path.getsize(svs) >> 20} MiB read through libtiff (nuhtc_amd.tiffslide), written in {t_svs:.1f} s (not timed)'} if args.svs else {})}))
    if world > 1:
        torch.distributed.destroy_process_group()
    if args.checkpoint is None and os.path.exists(ck):
        os.remove(ck)
    if args.svs and rank == 0:
        parallel.host_phase_cleanup('/tmp', rank, world)
        if os.path.exists(svs):
            os.remove(svs)


if __name__ == '__main__':
    main()
