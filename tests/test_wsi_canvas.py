"""BASELINE configs[2] (synthetic WSI): the canvas generator (CPU) and the slide-level path on the GPU -- tile stream,
detection records, cross-tile merge -- including the tile-sharded layout of the multi-GPU run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_pannuke_infer.py')


def test_canvas_is_rank_independent():
    """Any block of tile rows renders the same pixels as the whole canvas (each rank renders only its rows)."""
    from nuhtc_amd import synth
    full, y0 = synth.nuclei_canvas(5)
    assert y0 == 0 and full.shape == (synth.canvas_side(5),) * 2 + (3,) and full.dtype == np.uint8
    for rows in ((0, 1), (1, 4), (2, 3), (4, 5)):
        band, y = synth.nuclei_canvas(5, rows=rows)
        assert y == rows[0] * 192 and np.array_equal(band, full[y:y + band.shape[0]])
    par, _ = synth.nuclei_canvas_parallel(5, workers=3)
    assert np.array_equal(par, full)
    # H&E-like statistics: mostly background, a visible fraction of dark nuclei
    dark = (full.astype(np.int32).sum(-1) < 400).mean()
    assert 0.05 < dark < 0.5


def test_canvas_tiles_view():
    from nuhtc_amd import synth
    from nuhtc_amd.parallel import shard_range
    full, _ = synth.nuclei_canvas(3)
    lo, hi = shard_range(9, 1, 2)                      # second of two ranks: tiles 5..8, tile rows 1..2
    band, y0 = synth.nuclei_canvas(3, rows=(lo // 3, (hi - 1) // 3 + 1))
    t = synth.CanvasTiles(band, y0, 3, lo, hi)
    assert len(t) == hi - lo and t.shape == (hi - lo, 256, 256, 3)
    for k in range(len(t)):
        x, y = t.coords[k]
        assert (x, y) == ((lo + k) % 3 * 192, (lo + k) // 3 * 192)
        assert np.array_equal(t[k], full[y:y + 256, x:x + 256])
    assert t[1:3].shape == (2, 256, 256, 3) and np.array_equal(t[1:3][1], t[2])
    assert t[4:4].shape == (0, 256, 256, 3)


@pytest.fixture(scope='module')
def model(hip_device, tmp_path_factory):
    import torch
    from nuhtc_amd import weights
    from nuhtc_amd.apis import init_detector
    p = str(tmp_path_factory.mktemp('w') / 'w.pth')
    torch.save(dict(state_dict=weights.bench_state_dict(0)), p)
    m = init_detector(CFG, p, device='cuda:0', max_batch=8)
    m.opts.update(margin=2, min_area=10, mask_nms_thr=0.05)
    return m


@pytest.mark.gpu
def test_slide_merge_and_sharding(model):
    """4x4 tiles over one canvas: the device merge equals the sequential oracle on the slide's records, removes
    duplicates from the 64-pixel overlaps, and two 'ranks' (contiguous tile blocks, each rendering its own rows) produce
    the same merged slide as one."""
    from nuhtc_amd import synth, wsi
    from nuhtc_amd.parallel import shard_range
    from oracle.merge import merge_overlap as oracle_mask
    from oracle.merge_poly import merge_overlap_masks as oracle_poly
    G = 4
    full, _ = synth.nuclei_canvas(G)
    tiles = synth.CanvasTiles(full, 0, G, 0, G * G)
    rec = wsi.infer_tiles(model, tiles, tiles.coords, 8)
    n = len(rec['score'])
    assert n > 50
    kept = wsi.merge_overlap(rec, 0.05)                              # the reference's polygon-IoU measure (default)
    assert np.array_equal(kept, oracle_poly(rec['mask'], rec['score'], 0.05))
    kept_m = wsi.merge_overlap(rec, 0.05, overlap='mask')
    assert np.array_equal(kept_m, oracle_mask(rec, 0.05))
    print(f'{n} records: polygon-IoU keeps {len(kept)}, mask-IoU keeps {len(kept_m)}, they disagree on {len(np.setxor1d(kept, kept_m))}')
    # the rings the engine traced are the rings the oracle traces from the same masks
    from oracle import contour as OC
    for (m, x0, y0), ring in zip(rec['mask'], rec['ring']):
        assert np.array_equal(ring, OC.mask2inst(m) + np.array([x0, y0]))
    assert 0 < len(kept) < n                                         # overlap zones held duplicates
    parts = []
    for r in range(2):
        lo, hi = shard_range(G * G, r, 2)
        band, y0 = synth.nuclei_canvas(G, rows=(lo // G, (hi - 1) // G + 1))
        t = synth.CanvasTiles(band, y0, G, lo, hi)
        parts.append(wsi.infer_tiles(model, t, t.coords, 8))
    both = {k: parts[0][k] + parts[1][k] for k in ('score', 'mask', 'box', 'label')}
    assert len(both['score']) == n
    kept2 = wsi.merge_overlap(both, 0.05)
    # ragged batching (last batches of 1 and 2 tiles, fewer rows than the export capacity) must not change the records
    for bs in (5, 7):
        r2 = wsi.infer_tiles(model, tiles, tiles.coords, bs)
        assert r2['tile'] == rec['tile'] and r2['score'] == rec['score'] and all(np.array_equal(a, b) for a, b in zip(r2['ring'], rec['ring']))
    # more kept detections than the export buffers hold: the synchronous gather takes over, same records
    import nuhtc_amd.engine as E
    orig = E.Engine.export_async
    pipe = model.pipeline((256, 256), 3)
    try:
        E.Engine.export_async = lambda self, B, cap=None, contour_cap=256: orig(self, B, 8, contour_cap)
        r3 = wsi.infer_tiles(model, tiles, tiles.coords, 8)
    finally:
        E.Engine.export_async = orig
        for e in pipe.engines:
            e._ex = None
    assert r3['tile'] == rec['tile'] and r3['score'] == rec['score'] and all(np.array_equal(a[0], b[0]) for a, b in zip(r3['mask'], rec['mask']))
    key = lambda rc, i: (tuple(np.round(rc['box'][i], 3)), round(rc['score'][i], 6), rc['label'][i])
    assert sorted(key(rec, i) for i in kept) == sorted(key(both, i) for i in kept2)


@pytest.mark.gpu
def test_bench_wsi_cli(hip_device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools/bench_wsi.py'), '--grid', '6', '--workers', '2'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d['tiles'] == 36 and d['n_gpus'] == 1
    assert 0 < d['detections_after_merge'] < d['detections_after_tile_nms']
    assert d['tiles_per_s_inference'] > 0


@pytest.mark.gpu
def test_consep_1000px_crop_36_padded_tiles(hip_device, tmp_path):
    """BASELINE configs[3]: CoNSeP config (4 classes, max_per_img 300), a 1000x1000 crop -> tile origins np.arange(0,1000,192)
    = 6 per axis = 36 tiles of 256x256, the last row / column zero-padded past the edge (use_padding=True,
    tools/wsi_core/WholeSlideImage.py:419-421,460-466).  Oracle on a subset that includes edge-padded tiles and the padded
    corner; two tile-block shards ('2 x MI355X') give the same records and the same merged slide as one."""
    import torch
    import parity_util as P
    from nuhtc_amd import synth, weights, wsi
    from nuhtc_amd.apis import inference_detector, init_detector
    from nuhtc_amd.parallel import shard_range
    from oracle import model as O
    from oracle.merge_poly import merge_overlap_masks as oracle_poly
    cfg = os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_consep_infer.py')
    sd = weights.bench_state_dict(5, num_classes=4, obj_bias=0.3)
    # these 4-class synthetic weights paint a few scattered pixels per box (rings that enclose no area, so the polygon measure of the
    # cross-tile merge would have nothing to remove); a positive mask-logit bias turns them into blobs that fill most of their box
    # (oracle: mean 159 px, two thirds of the rings enclose > 4 px^2)
    sd['roi_head.mask_head.0.conv_logits.bias'] = sd['roi_head.mask_head.0.conv_logits.bias'] + 8.0
    ck = str(tmp_path / 'consep.pth')
    torch.save(dict(state_dict=sd), ck)
    model = init_detector(cfg, ck, device='cuda:0', max_batch=12)
    assert model.opts['num_classes'] == 4 and model.opts['max_per_img'] == 300
    model.opts.update(margin=2, min_area=10, mask_nms_thr=0.05)
    # a 1000x1000 crop: four 512-pixel synthetic fields, cut to size
    img = np.concatenate([np.concatenate([synth.nuclei_tile(300 + 2 * r + c, 512, mean_count=60) for c in range(2)], 1) for r in range(2)], 0)[:1000, :1000]
    tiles, coords = wsi.tile_grid(img, 256, 192)
    assert tiles.shape == (36, 256, 256, 3) and coords[-1].tolist() == [960, 960]
    assert tiles[5][:, 40:].max() == 0 and tiles[35][40:, :].max() == 0 and tiles[35][:40, :40].max() > 0      # 1000 - 960 = 40 px of image
    # --- engine vs oracle on interior, edge-padded and corner tiles (the API path of tools/infer_wsi.py: ndarray input)
    sub = [0, 5, 14, 30, 35]
    got = inference_detector(model, [tiles[i] for i in sub])
    ref, it = O.Oracle(sd, num_classes=4, max_per_img=300)(tiles[sub], 1, keep=True)
    vals = P.oracle_paste_values(O, it, (256, 256))
    for k, i in enumerate(sub):
        rep, fails = P.compare_strict(ref[k], got[k], max_per_img=300, values=vals[k])
        print(f'consep crop tile {i} (origin {coords[i].tolist()}): {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        assert not fails, (i, fails)
        assert len(got[k][0]) == 4
    # --- the whole crop, one rank vs two contiguous shards
    rec = wsi.infer_tiles(model, tiles, coords, 12)
    n = len(rec['score'])
    assert n > 100 and max(rec['label']) <= 3
    parts = []
    for r in range(2):
        lo, hi = shard_range(36, r, 2)
        p = wsi.infer_tiles(model, tiles[lo:hi], coords[lo:hi], 12)
        p['tile'] = [t + lo for t in p['tile']]
        parts.append(p)
    both = {k: parts[0][k] + parts[1][k] for k in rec}
    assert both['tile'] == rec['tile'] and both['score'] == rec['score'] and all(np.array_equal(a, b) for a, b in zip(both['ring'], rec['ring']))
    kept = wsi.merge_overlap(rec, 0.05)
    assert np.array_equal(kept, oracle_poly(rec['mask'], rec['score'], 0.05))
    # the 64-pixel tile overlaps produce duplicates, and with area-enclosing rings the reference's polygon measure removes them
    assert np.array_equal(kept, wsi.merge_overlap(both, 0.05)) and 0 < len(kept) < n
    kept_m = wsi.merge_overlap(rec, 0.05, overlap='mask')
    assert np.array_equal(kept_m, wsi.merge_overlap(both, 0.05, overlap='mask')) and 0 < len(kept_m) < n
    # no detection reaches into the zero padding beyond the 2-pixel margin rule, and masks stay inside the crop + tile frame
    for (m, x0, y0) in rec['mask']:
        assert x0 >= 0 and y0 >= 0 and x0 + m.shape[1] <= 960 + 256 and y0 + m.shape[0] <= 960 + 256
    print(f'consep crop: {n} nuclei after per-tile mask-NMS, {len(kept)} after the cross-tile merge (polygon IoU), {len(kept_m)} (mask IoU)')


@pytest.mark.gpu
def test_device_crops_equal_per_detection_unpack(model):
    """The slide loop's array path (masks cropped on the device by nuhtc_export_crops, records built with whole-batch array
    operations) against the per-detection path (full masks fetched, cropped one by one on the host): same records in the same
    order -- tile, box, score, label, mask crop and origin, closed ring -- and the same packed gather tensors and merge result."""
    import torch
    from nuhtc_amd import hip, synth, wsi
    G = 4
    full, _ = synth.nuclei_canvas(G)
    tiles = synth.CanvasTiles(full, 0, G, 0, G * G)
    rec = wsi.infer_tiles(model, tiles, tiles.coords, 8)
    assert isinstance(rec['mask'], wsi.PackedMasks) and len(rec['tile']) > 50
    ref = dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
    pipe = model.pipeline(tiles.shape[1:3], 4)

    def finish():
        eng, B, stream, i0 = pipe.collect()
        with torch.cuda.stream(stream):
            wsi._unpack(eng, B, i0, tiles.coords, 256, ref, exported=False)
    for i in range(0, len(tiles), 8):
        if pipe.full():
            finish()
        pipe.submit(tiles[i:i + 8], hip.CH_SWAP, tag=i, export=True)
    while pipe.pending:
        finish()
    assert rec['tile'] == ref['tile'] and rec['score'] == ref['score'] and rec['label'] == ref['label']
    assert all(np.array_equal(a, b) for a, b in zip(rec['box'], ref['box']))
    assert all(np.array_equal(a, b) for a, b in zip(rec['ring'], ref['ring']))
    for (m, x0, y0), (m2, x2, y2) in zip(rec['mask'], ref['mask']):
        assert (x0, y0) == (x2, y2) and np.array_equal(m, m2)
    for keep in (None, list(range(0, len(ref['tile']), 3))):
        for a, b in zip(wsi.pack_records(rec, keep), wsi.pack_records(ref, keep)):
            assert a.shape == b.shape and bool((a == b).all())
    assert np.array_equal(wsi.merge_overlap(rec, 0.05), wsi.merge_overlap(ref, 0.05))
