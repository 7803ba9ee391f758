"""GPU tests of the reference-mirroring Python API (init_detector / inference_detector) and the WSI host path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_pannuke_infer.py')


@pytest.fixture(scope='module')
def model(tmp_path_factory):
    import torch
    from nuhtc_amd import weights
    from nuhtc_amd.apis import init_detector
    p = str(tmp_path_factory.mktemp('ck') / 'synthetic.pth')
    torch.save(dict(meta={}, state_dict=weights.bench_state_dict(0, obj_bias=0.0)), p)
    return init_detector(CFG, p, device='cuda:0', max_batch=4)


def test_inference_detector_formats_and_channel_modes(hip_device, model, tmp_path):
    from PIL import Image
    from nuhtc_amd import synth
    from nuhtc_amd.apis import inference_detector
    from oracle import model as O
    tiles = synth.nuclei_tiles(3, 64, start=5)
    # list of ndarrays: batch result, WSI channel handling (array treated as BGR and swapped)
    res = inference_detector(model, [t for t in tiles])
    assert isinstance(res, list) and len(res) == 3
    import parity_util as P
    ref, it = O.Oracle(model.state_dict)(tiles, 1, keep=True)
    vals = P.oracle_paste_values(O, it, (64, 64))
    for i, ((gb, gm), (rb, rm)) in enumerate(zip(res, ref)):
        assert len(gb) == 5 and len(gm) == 5
        rep, fails = P.compare_strict((rb, rm), (gb, gm), values=vals[i])
        print(f'tile {i}: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        assert not fails, fails
        for c in range(5):
            assert gb[c].dtype == np.float32 and len(gm[c]) == len(gb[c])
            if len(gb[c]) and gb[c].shape == rb[c].shape:
                assert np.abs(gb[c] - rb[c]).max() < 1e-4     # rows in the same (NMS) order, boxes to 1e-4 px
                assert all(m.dtype == bool and m.shape == (64, 64) for m in gm[c])
    # single ndarray: single result
    one = inference_detector(model, tiles[0])
    assert isinstance(one, tuple) and all(np.array_equal(a, b) for a, b in zip(one[0], res[0][0]))
    # file path: tools/infer.py handling (true RGB meets RGB means)
    p = str(tmp_path / 't.png')
    Image.fromarray(tiles[0]).save(p)
    f = inference_detector(model, p)
    rf = O.Oracle(model.state_dict)(tiles[:1], 0)[0]
    assert [len(b) for b in f[0]] == [len(b) for b in rf[0]]
    # more images than max_batch are chunked
    many = inference_detector(model, [tiles[i % 3] for i in range(6)])
    assert len(many) == 6 and all(np.array_equal(a, b) for a, b in zip(many[3][0], many[0][0]))
    # a list of mixed sizes (mmdet/apis/inference.py:118-139 accepts one): grouped by size, results in input order, each image as in a
    # call of its own
    small = synth.nuclei_tiles(2, 32, start=90)
    mixed = inference_detector(model, [tiles[0], small[0], tiles[1], small[1]])
    alone = [inference_detector(model, t) for t in (tiles[0], small[0], tiles[1], small[1])]
    assert len(mixed) == 4
    for m, a in zip(mixed, alone):
        assert all(np.array_equal(x, y) for x, y in zip(m[0], a[0]))
        assert all(len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y)) for x, y in zip(m[1], a[1]))
    assert mixed[1][1][0] == [] or mixed[1][1][0][0].shape == (32, 32)
    with pytest.raises(ValueError):
        inference_detector(model, [tiles[0], np.zeros((32, 32), np.uint8)])


def test_wsi_host_path_single_rank(hip_device, model):
    from nuhtc_amd import synth, wsi
    rng_img = np.concatenate([np.concatenate([synth.nuclei_tile(10 + 2 * r + c, 128) for c in range(2)], 1) for r in range(2)], 0)
    tiles, coords = wsi.tile_grid(rng_img, 64, 48)
    rec = wsi.infer_tiles(model, tiles, coords, batch_size=4)
    n = len(rec['score'])
    assert n > 0 and len(rec['mask']) == n == len(rec['ring'])
    from nuhtc_amd import contours
    for (m, x0, y0), ring in zip(rec['mask'], rec['ring']):      # GPU-traced ring == host trace of the same mask
        assert np.array_equal(ring, contours.mask_to_ring(m, origin=(x0, y0)))
    keep = wsi.merge_overlap(rec, 0.05)
    assert 0 < len(keep) <= n
    # kept detections do not overlap each other above the threshold
    sub = {k: [rec[k][i] for i in keep] for k in rec}
    assert len(wsi.merge_overlap(sub, 0.05)) == len(keep)


def test_infer_wsi_cli_writes_qupath_geojson(hip_device, tmp_path):
    import json
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, weights
    img = np.concatenate([np.concatenate([synth.nuclei_tile(30 + 2 * r + c, 128) for c in range(2)], 1) for r in range(2)], 0)
    src = tmp_path / 'slide.npy'
    np.save(src, img)
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools/infer_wsi.py'), str(src), CFG, str(ck), '--patch_size', '64', '--step_size', '48',
                           '--batch_size', '4', '--margin', '2', '--save_dir', str(tmp_path / 'out'), '--merge', '--mode', 'all'])
    feats = json.load(open(tmp_path / 'out/nuclei/slide/slide.geojson'))
    merged = json.load(open(tmp_path / 'out/nuclei/slide/slide_merged.geojson'))
    pts = json.load(open(tmp_path / 'out/nuclei/slide/slide_point.geojson'))
    assert len(feats) == len(pts) > 0 and 0 < len(merged) <= len(feats)
    f0 = feats[0]
    assert f0['geometry']['type'] == 'Polygon' and f0['geometry']['coordinates'][0][0] == f0['geometry']['coordinates'][0][-1]
    assert set(f0['properties']) >= {'objectType', 'label', 'score', 'classification', 'isLocked'}
    # the other output modes describe the same detections
    import sqlite3
    from nuhtc_amd import cocomask
    dsa = json.load(open(tmp_path / 'out/nuclei/slide/slide_dsa.json'))
    assert len(dsa['elements']) == len(feats) and dsa['elements'][0]['points'][0][:2] == f0['geometry']['coordinates'][0][0]
    coco = json.load(open(tmp_path / 'out/nuclei/slide/coco_nuclei.json'))
    assert len(coco['annotations']) == len(feats) and sum(i['n_objects'] for i in coco['images']) == len(feats)
    a0 = coco['annotations'][0]
    m0 = cocomask.decode(a0['segmentation'])
    assert m0.shape == (64, 64) and m0.sum() >= 10 and os.path.exists(tmp_path / 'out/imgs/slide' / f"{a0['image_id']}.png")
    n_sql = sqlite3.connect(str(tmp_path / 'out/nuclei/slide/slide_dql.db')).execute('SELECT COUNT(*) FROM contour').fetchone()[0]
    assert n_sql == len(feats)
    # ---- every document against the reference's own statements: the reference's per-tile loop (tools/infer_wsi.py:486-546)
    # and its writers (:549-693) restated in tests/ref_consume.py + oracle/writers.py + oracle/rle.py + oracle/contour.py, fed with what
    # `inference_detector` returns for the same tiles from an engine of this process; object for object, key order included
    import ref_consume as RC
    from oracle import writers as W
    from nuhtc_amd import wsi
    from nuhtc_amd.apis import inference_detector, init_detector
    tiles, coords = wsi.tile_grid(img, 64, 48)
    det = init_detector(CFG, str(ck), device='cuda:0', max_batch=4)
    results = inference_detector(det, [t for t in tiles])
    ref, kept = RC.wsi_documents(results, coords, 64, margin=2, min_area=10)
    rt = lambda x: json.loads(json.dumps(x))
    assert len(ref['geojson']) == len(feats) > 20
    assert feats == rt(ref['geojson']) and list(feats[0]) == list(ref['geojson'][0]) and list(feats[0]['properties']) == list(ref['geojson'][0]['properties'])
    assert pts == rt(ref['pointjson'])
    assert dsa == rt(W.dsa_file(ref['dsajson'])) and list(dsa['elements'][0]) == list(ref['dsajson'][0])
    assert coco == rt(W.coco_file(ref['imgs'], ref['annts']))
    rows = sqlite3.connect(str(tmp_path / 'out/nuclei/slide/slide_dql.db')).execute(
        'SELECT annidx, elementidx, type, "group", score, color, xmin, ymin, xmax, ymax, bbox_area, coords_x, coords_y, keep FROM contour ORDER BY id').fetchall()
    assert rows == ref['sql_rows']
    # the tile images of --mode coco are the tiles themselves
    from PIL import Image
    for im in coco['images'][:3]:
        assert np.array_equal(np.asarray(Image.open(tmp_path / 'out/imgs/slide' / im['file_name'])), tiles[im['id']])
    # and the merged file holds exactly the features the sequential polygon merge of the reference keeps (oracle/merge_poly.py)
    from oracle import merge_poly as MP
    keep_ref = MP.merge_overlap([k[1] for k in kept], [k[3] for k in kept], 0.05)
    assert 0 < len(keep_ref) <= len(feats) and merged == [feats[i] for i in keep_ref]


def test_infer_wsi_tissue_coords_and_store_front_ends(hip_device, tmp_path):
    """SURVEY 8f row 2 on the GPU box: `tools/infer_wsi.py --seg` (segmentTissue + process_contours, WholeSlideImage.py:105-199,388-502),
    `--coords file.npz` (the role of patches/<name>.h5) and a store directory (Whole_Slide_Bag_FP, :832-898) on a synthetic slide with two
    tissue regions, a hole and a speck.  The tile list is derived independently by oracle/tissue.py (the reference's sequence over the
    scalar OpenCV restatements of oracle/cv_ops.py); the tiles are cut here by plain slicing and handed over as .npz: every front end must
    write the SAME GeoJSON as that run, i.e. it used exactly the oracle's tiles, in the oracle's order, with the same pixels."""
    import json
    import subprocess
    import sys
    import torch
    from nuhtc_amd import tilestore, weights
    from oracle import tissue as OT
    from test_tissue import tissue_slide_with_holes
    img, blob1, hole, blob2 = tissue_slide_with_holes()
    P = 64
    regions = OT.segment_tissue(img, 8)
    assert len(regions) == 2 and sorted(len(hs) for _, hs in regions) == [0, 1]
    # regions in cv2's RETR_CCOMP list order (restated in oracle/contour.py: newest border first)
    per_contour = [OT.contour_tile_coords(c, hs, P, P) for c, hs in regions]
    assert all(len(pts) > 50 for pts in per_contour)
    coords = np.array([p for pts in per_contour for p in pts], np.int64)
    H, W = img.shape[:2]
    tiles = np.zeros((len(coords), P, P, 3), np.uint8)
    for i, (x, y) in enumerate(coords):
        sub = img[y:min(y + P, H), x:min(x + P, W)]
        tiles[i, :sub.shape[0], :sub.shape[1]] = sub
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    np.save(tmp_path / 'slide.npy', img)
    np.savez(tmp_path / 'given.npz', tiles=tiles, coords=coords)
    np.savez(tmp_path / 'coords.npz', coords=coords, patch_size=P)
    tilestore.write_store(str(tmp_path / 'stored'), img, coords, patch_size=P)
    tool = os.path.join(ROOT, 'tools/infer_wsi.py')
    common = [CFG, str(ck), '--patch_size', str(P), '--step_size', str(P), '--batch_size', '16']

    def run(name, src, *extra):
        out = subprocess.run([sys.executable, tool, str(src)] + common + ['--save_dir', str(tmp_path / name)] + list(extra), check=True,
                             capture_output=True, text=True).stdout
        stem = os.path.splitext(os.path.basename(str(src)))[0]
        return json.load(open(tmp_path / name / 'nuclei' / stem / f'{stem}.geojson')), out

    ref, _ = run('o_npz', tmp_path / 'given.npz')
    assert len(ref) > 300
    seg, log = run('o_seg', tmp_path / 'slide.npy', '--seg', '--seg_downsample', '8')
    assert f'2 contour(s), {len(coords)} tiles' in log
    assert seg == ref
    assert run('o_coords', tmp_path / 'slide.npy', '--coords', str(tmp_path / 'coords.npz'))[0] == ref
    assert run('o_store', tmp_path / 'stored')[0] == ref
    # nothing was detected on glass or inside the hole: every ring lies in a tile the oracle listed
    xs = np.array([f['geometry']['coordinates'][0][0] for f in ref])
    assert not hole[np.clip(xs[:, 1], 0, H - 1), np.clip(xs[:, 0], 0, W - 1)].all()


def test_readme_command_lines_on_a_folder_of_slides(hip_device, tmp_path):
    """The reference's documented invocations (README.md:55-63 and :221-223), unchanged except for the paths and `--slide_ext .npy`:
    a FOLDER of slides -> process_list_autogen.csv, masks/, patches/, stitches/, nuclei/<id>/<id>.geojson per slide; each slide's GeoJSON
    equals the single-slide run of the same flags; a slide with a merged file is skipped on the next run (tools/infer_wsi.py:456-458);
    --det writes the per-tile overlays (:504-512)."""
    import json
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, weights
    H = W = 3072
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    blob = ((yy - 1500) / 1250.0) ** 2 + ((xx - 1600) / 1300.0) ** 2 <= 1
    tex = np.concatenate([np.concatenate([synth.nuclei_tile(900 + 12 * r + c, 256) for c in range(W // 256)], 1) for r in range(H // 256)], 0)
    img = np.full((H, W, 3), 235, np.uint8)
    img[blob] = tex[blob]
    src = tmp_path / 'wsi'
    src.mkdir()
    np.save(src / 'a.npy', img)
    np.save(src / 'b.npy', np.ascontiguousarray(img[::-1, ::-1]))
    ck = tmp_path / 'pannuke.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0)), ck)
    tool = os.path.join(ROOT, 'tools/infer_wsi.py')
    out = tmp_path / 'wsi_infer'
    # README.md:55-63
    line1 = f'{src} {CFG} {ck} --patch --seg --stitch --patch_size 256 --step_size 192 --batch_size 16 --save_dir {out} --mode qupath --slide_ext .npy'
    log = subprocess.run([sys.executable, tool] + line1.split(), check=True, capture_output=True, text=True).stdout
    rows = open(out / 'process_list_autogen.csv').read().splitlines()
    assert rows[0].startswith('slide_id,process,status,seg_level') and rows[1].startswith('a.npy,0,processed,6,') and rows[2].startswith('b.npy,0,processed,6,')
    docs = {}
    for sid in ('a', 'b'):
        for f in (out / 'masks' / f'{sid}.png', out / 'patches' / f'{sid}.npz', out / 'stitches' / f'{sid}.jpg', out / 'nuclei' / sid / f'{sid}_point.geojson'):
            assert os.path.exists(f), f
        docs[sid] = json.load(open(out / 'nuclei' / sid / f'{sid}.geojson'))
        assert len(docs[sid]) > 1000
        n_tiles = len(np.load(out / 'patches' / f'{sid}.npz')['coords'])
        assert 100 < n_tiles < 260 and f'{n_tiles} tiles on 1 rank(s)' in log
    # the single-slide form of the same flags gives the same documents
    one = tmp_path / 'one'
    subprocess.run([sys.executable, tool, str(src / 'b.npy')] + line1.split()[1:-6] + ['--save_dir', str(one), '--mode', 'qupath'], check=True, capture_output=True)
    assert json.load(open(one / 'nuclei/b/b.geojson')) == docs['b']
    # the same folder on two ranks (`--gpus 2`: the tool starts them itself; rank 0 segments and patches, both shard every slide's tiles,
    # one gather per slide): the same documents, byte for byte
    env2 = _two_rank_env()
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env2.pop(k, None)
    two = tmp_path / 'two_ranks'
    r2 = subprocess.run([sys.executable, tool] + line1.split()[:-6] + ['--save_dir', str(two), '--mode', 'qupath', '--slide_ext', '.npy', '--gpus', '2'], env=env2,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(r2.stdout[-1500:])
    assert r2.returncode == 0 and 'tiles on 2 rank(s)' in r2.stdout
    for sid in ('a', 'b'):
        assert open(two / 'nuclei' / sid / f'{sid}.geojson', 'rb').read() == open(out / 'nuclei' / sid / f'{sid}.geojson', 'rb').read()
        assert open(two / 'patches' / f'{sid}.npz', 'rb').read() == open(out / 'patches' / f'{sid}.npz', 'rb').read()
    # a patch folder as the REFERENCE's tool leaves it -- only patches/<id>.h5 (WholeSlideImage.process_contours) --: the slides count as patched
    # (auto-skip, tools/infer_wsi.py:168) and the slide loop reads the .h5 (Whole_Slide_Bag_FP): the same documents
    from nuhtc_amd import h5coords
    if h5coords.available():
        import shutil
        ref_made = tmp_path / 'ref_made'
        os.makedirs(ref_made / 'patches')
        for sid in ('a', 'b'):
            assert os.path.exists(out / 'patches' / f'{sid}.h5')
            shutil.copy(out / 'patches' / f'{sid}.h5', ref_made / 'patches' / f'{sid}.h5')
        log_h5 = subprocess.run([sys.executable, tool] + line1.split()[:-6] + ['--save_dir', str(ref_made), '--mode', 'qupath', '--slide_ext', '.npy'], check=True,
                                capture_output=True, text=True).stdout
        assert 'a already exist in destination location, skipped' in log_h5 and not os.path.exists(ref_made / 'patches' / 'a.npz')
        for sid in ('a', 'b'):
            assert json.load(open(ref_made / 'nuclei' / sid / f'{sid}.geojson')) == docs[sid]
    # README.md:221-223 into the same directory: the coordinate files exist (auto-skip), both slides are inferred again with margin 1
    line2 = f'{src} {CFG} {ck} --patch --seg --stitch --patch_size 256 --step_size 192 --margin 1 --min_area 10 --batch_size 32 --save_dir {out} --mode qupath --slide_ext .npy'
    log2 = subprocess.run([sys.executable, tool] + line2.split() + ['--merge', '--det', '--score-thr', '0.5'], check=True, capture_output=True, text=True).stdout
    assert 'a already exist in destination location, skipped' in log2
    m1 = json.load(open(out / 'nuclei/a/a.geojson'))
    assert 0 < len(m1) < len(docs['a'])                                  # margin 1 drops the boxes that touch the tile edge
    merged = json.load(open(out / 'nuclei/a/a_merged.geojson'))
    assert 0 < len(merged) < len(m1)
    dets = os.listdir(out / 'a' / 'infer')
    xs = np.load(out / 'patches' / 'a.npz')['coords']
    assert len(dets) > 50 and all(d.startswith('img_') and d.endswith('.jpg') for d in dets) and f'img_{xs[0][0]}_{xs[0][1]}.jpg' in dets
    # third run: both slides have their merged file -> skipped
    log3 = subprocess.run([sys.executable, tool] + line2.split(), check=True, capture_output=True, text=True).stdout
    assert 'skip a due to existing results' in log3 and 'skip b due to existing results' in log3 and 'rank(s)' not in log3
    # wrong extension: slide ids keep their suffix and no coordinate file matches (the reference's behaviour for a wrong --slide_ext)
    log4 = subprocess.run([sys.executable, tool] + line2.split()[:-1] + ['.svs'], check=True, capture_output=True, text=True).stdout
    assert 'skip a.npy due to no coord file' in log4


def test_readme_command_line_verbatim_on_svs_slides(hip_device, tmp_path):
    """README.md:55-63 with NOTHING changed but the paths: the slides are Aperio-layout `.svs` files (tiled pyramidal TIFF, the reference's
    default --slide_ext), opened through libtiff (nuhtc_amd.tiffslide) where the reference uses OpenSlide.  The file is written losslessly, so
    its level-0 tiles are the source array's pixels: the same array as a `.npy` slide with the coordinate file of the `.svs` run gives the same
    GeoJSON, byte for byte."""
    import json
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, tiffslide, weights
    if not tiffslide.available():
        pytest.skip('libtiff not found')
    H = W = 3072
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    blob = ((yy - 1500) / 1250.0) ** 2 + ((xx - 1600) / 1300.0) ** 2 <= 1
    tex = np.concatenate([np.concatenate([synth.nuclei_tile(900 + 12 * r + c, 256) for c in range(W // 256)], 1) for r in range(H // 256)], 0)
    img = np.full((H, W, 3), 235, np.uint8)
    img[blob] = tex[blob]
    src = tmp_path / 'wsi'
    src.mkdir()
    tiffslide.write_pyramid(str(src / 'TCGA-01.svs'), img, levels=4, tile=240, compression='lzw',
                            description='Aperio Image Library v11.2.1\n3072x3072 (240x240) JPEG/RGB Q=30|AppMag = 40|MPP = 0.2520')
    ck = tmp_path / 'pannuke.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0)), ck)
    tool = os.path.join(ROOT, 'tools/infer_wsi.py')
    out = tmp_path / 'wsi_infer'
    line = f'{src} {CFG} {ck} --patch --seg --stitch --patch_size 256 --step_size 192 --batch_size 16 --save_dir {out} --mode qupath'
    log = subprocess.run([sys.executable, tool] + line.split(), check=True, capture_output=True, text=True).stdout
    rows = open(out / 'process_list_autogen.csv').read().splitlines()
    assert rows[1].startswith('TCGA-01.svs,0,processed,3,')          # seg_level -1 -> the FILE's level nearest 64x: level 3 (8x) of this 1 / 2 / 4 / 8 pyramid
    for f in (out / 'masks' / 'TCGA-01.png', out / 'patches' / 'TCGA-01.npz', out / 'stitches' / 'TCGA-01.jpg', out / 'nuclei' / 'TCGA-01' / 'TCGA-01_point.geojson'):
        assert os.path.exists(f), f
    doc = open(out / 'nuclei' / 'TCGA-01' / 'TCGA-01.geojson', 'rb').read()
    n_tiles = len(np.load(out / 'patches' / 'TCGA-01.npz')['coords'])
    assert len(json.loads(doc)) > 1000 and 100 < n_tiles < 260 and f'{n_tiles} tiles on 1 rank(s)' in log
    np.save(tmp_path / 'TCGA-01.npy', img)
    one = tmp_path / 'one'
    subprocess.run([sys.executable, tool, str(tmp_path / 'TCGA-01.npy'), CFG, str(ck), '--patch_size', '256', '--step_size', '192', '--batch_size', '16', '--save_dir', str(one),
                    '--mode', 'qupath', '--coords', str(out / 'patches' / 'TCGA-01.npz')], check=True, capture_output=True)
    assert open(one / 'nuclei' / 'TCGA-01' / 'TCGA-01.geojson', 'rb').read() == doc


def test_infer_patch_cli_writes_coco(hip_device, tmp_path):
    import json
    import subprocess
    import sys
    import torch
    from PIL import Image
    from nuhtc_amd import cocomask, synth, weights
    tiles = synth.nuclei_tiles(5, 64, start=60)
    with open(tmp_path / 'labels.csv', 'w') as f:
        f.write('image_path,other\n')
        for i, t in enumerate(tiles):
            Image.fromarray(t).save(tmp_path / f'im{i}.png')
            f.write(f"{tmp_path / f'im{i}.png'},x\n")
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools/infer_patch.py'), '--csv', str(tmp_path / 'labels.csv'), '--config', CFG,
                           '--checkpoint', str(ck), '--output', str(tmp_path / 'o/nuclei_coco.json'), '--batch-size', '4'])
    doc = json.load(open(tmp_path / 'o/nuclei_coco.json'))
    assert [im['id'] for im in doc['images']] == [1, 2, 3, 4, 5] and doc['images'][0]['height'] == 64
    assert len(doc['annotations']) > 0 and doc['categories'][0]['name'] == 'nucleus'
    ids = [a['id'] for a in doc['annotations']]
    assert ids == list(range(len(ids)))
    for a in doc['annotations'][:5]:
        m = cocomask.decode(a['segmentation'])
        ys, xs = np.nonzero(m)
        assert a['bbox'] == [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1] and 0.35 <= a['score'] <= 1
    # per image, kept instances do not overlap above the mask-NMS threshold
    by_img = {}
    for a in doc['annotations']:
        by_img.setdefault(a['image_id'], []).append(a['segmentation'])
    for rl in by_img.values():
        iou = cocomask.iou(rl, rl)
        assert (iou - np.eye(len(rl))).max() <= 0.05
    # ---- the annotations against the reference's statements (tools/infer_patch.py:247-290 restated: tests/ref_consume.py,
    # oracle/writers.py, oracle/rle.py) on what inference_detector returns for the same images in this process: object for object
    import ref_consume as RC
    from nuhtc_amd.apis import inference_detector, init_detector
    det = init_detector(CFG, str(ck), device='cuda:0', max_batch=4)
    results = inference_detector(det, [np.array(Image.open(tmp_path / f'im{i}.png').convert('RGB')) for i in range(5)])
    want = RC.patch_annotations(results, first_image_id=1, thr=0.05)
    assert len(want) == len(doc['annotations']) > 10 and doc['annotations'] == json.loads(json.dumps(want))
    assert list(doc['annotations'][0]) == list(want[0])
    assert doc['images'] == [{'id': i + 1, 'file_name': f'im{i}.png', 'img_path': str(tmp_path / f'im{i}.png'), 'height': 64, 'width': 64} for i in range(5)]


def test_pannuke_dataset_cli_exports_and_scores(hip_device, tmp_path):
    """tools/test_pannuke.py: export a fold in the PanNuke array format, then score a second run against that export:
    the engine is deterministic, so every metric of the second run is 1 (PQ up to its 1e-6 epsilon)."""
    import json
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, weights
    imgs = synth.nuclei_tiles(5, 64, start=40)
    np.save(tmp_path / 'images.npy', imgs)
    np.save(tmp_path / 'types.npy', np.array(['Breast', 'Colon', 'Breast', 'Lung', 'Colon']))
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    tool = os.path.join(ROOT, 'tools/test_pannuke.py')
    subprocess.check_call([sys.executable, tool, CFG, str(ck), '--images', str(tmp_path / 'images.npy'), '--out', str(tmp_path / 'a'),
                           '--batch', '4'])
    pred = np.load(tmp_path / 'a' / 'preds_pannuke.npy')
    assert pred.shape == (5, 64, 64, 6) and pred[..., :5].max() > 0
    assert np.array_equal(pred[..., 5], 1 - (pred[..., :5].max(-1) > 0))
    subprocess.check_call([sys.executable, tool, CFG, str(ck), '--images', str(tmp_path / 'images.npy'), '--masks',
                           str(tmp_path / 'a' / 'preds_pannuke.npy'), '--types', str(tmp_path / 'types.npy'), '--out', str(tmp_path / 'b'),
                           '--batch', '4'])
    s = json.load(open(tmp_path / 'b' / 'summary.json'))
    # instances hidden under a later instance of the same class are absent from the export, so detection quality can
    # drop slightly below 1 while every surviving pair matches exactly
    assert s['sq'] > 0.97 and s['dq'] > 0.9 and s['bPQ'] > 0.85 and max(s[f'multi_pq+_{c}'] for c in range(5)) > 0.85
    assert os.path.exists(tmp_path / 'b' / 'class_stats.csv') and os.path.exists(tmp_path / 'b' / 'tissue_stats.csv')


class _OracleContour:
    """The checker of the contour kernel: oracle/contour.py, the restatement of the published Suzuki-Abe border following with
    OpenCV's list order / start pixel / orientation / CHAIN_APPROX_SIMPLE rule (not the product's own host tracer)."""

    @staticmethod
    def trace_outer_contour(mask):
        from oracle import contour as OC
        c, _ = OC.find_contours_tree(mask)
        return c[0] if c else np.zeros((0, 2), np.int64)


def test_device_contours_match_oracle(hip_device, model):
    """nuhtc_mask_contours against oracle/contour.py (`cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]`), vertex by
    vertex (integer work: exact), on the engine's own masks and on hand-made shapes written into the mask buffer (thin lines,
    holes, single pixels, spurs, several blobs, an island inside a hole, blobs that touch the tile border, random fragmented
    masks, a contour longer than the device capacity)."""
    import torch
    from nuhtc_amd import synth
    host = _OracleContour
    eng = model.engine((64, 64))
    tiles = synth.nuclei_tiles(4, 64, start=70)
    B = eng.infer_async(eng.to_device(tiles), 1)
    eng.check()
    got = eng.contours(B, cap=256, kept_only=True)
    keep = eng.keep[:B].cpu().numpy()
    counts = eng.counts[:B].cpu().numpy()
    n_checked = 0
    for b in range(B):
        assert sorted(got[b]) == [int(i) for i in np.nonzero(keep[b, :counts[b]])[0]]
        for sl, ring in got[b].items():
            words = eng.masks[b, sl].cpu().numpy().view(np.uint32)
            bits = np.unpackbits(words.view(np.uint8).reshape(64, 8), axis=-1, bitorder='little').astype(bool)
            assert np.array_equal(ring, host.trace_outer_contour(bits))
            n_checked += 1
    assert n_checked > 10
    # hand-made masks in slots 0.. of tile 0
    shapes = []
    m = np.zeros((64, 64), bool); m[10, 5:40] = True; shapes.append(m)                       # horizontal line
    m = np.zeros((64, 64), bool); m[5:50, 7] = True; shapes.append(m)                        # vertical line
    m = np.zeros((64, 64), bool); m[20, 20] = True; shapes.append(m)                         # single pixel
    m = np.zeros((64, 64), bool); m[20, 20] = m[21, 21] = True; shapes.append(m)             # two diagonal pixels
    m = np.zeros((64, 64), bool); m[8:30, 8:30] = True; m[14:20, 14:20] = False; shapes.append(m)   # hole
    m = np.zeros((64, 64), bool); m[0:12, 0:9] = True; m[55:64, 50:64] = True; shapes.append(m)      # two blobs on the border
    m = np.ones((64, 64), bool); shapes.append(m)                                            # full tile
    yy, xx = np.mgrid[0:64, 0:64]
    m = ((yy - 30) ** 2 / 400 + (xx - 28) ** 2 / 150) <= 1; shapes.append(m)                 # ellipse
    m = (xx + yy) % 2 == 0; m[:, 40:] = False; shapes.append(m)                              # checkerboard: long 8-connected border
    rng = np.random.default_rng(3)
    m = rng.uniform(size=(64, 64)) < 0.55; shapes.append(m)                                  # noise
    m = np.zeros((64, 64), bool); m[3:9, 3:9] = True; m[30:40, 10:20] = True; m[30:35, 40:50] = True; shapes.append(m)   # three blobs: [0][0] is the last found
    m = np.zeros((64, 64), bool); m[5:40, 5:40] = True; m[10:35, 10:35] = False; m[20:25, 20:25] = True; shapes.append(m)  # island inside a hole: not top-level
    m = m.copy(); m[50:55, 2:8] = True; shapes.append(m)                                     # ... plus a later top-level blob
    m = np.zeros((64, 64), bool); m[10:13, 10:13] = True; m[11, 13:20] = True; shapes.append(m)     # block with a one-pixel spur
    m = np.zeros((64, 64), bool); m[5:30, 5] = True; m[29, 5:30] = True; shapes.append(m)    # one-pixel L
    m = np.zeros((64, 64), bool); m[20:30, 0:6] = True; m[20:24, 58:64] = True; shapes.append(m)    # blobs on the left / right edges, same first row
    from scipy import ndimage as ndi
    for k in range(24):                                                                      # smooth random blobs, fragmented
        a = ndi.gaussian_filter(rng.standard_normal((64, 64)), rng.uniform(1.5, 4.0))
        m = a > np.quantile(a, rng.uniform(0.6, 0.9))
        if k % 3 == 0:
            m &= ~(ndi.gaussian_filter(rng.standard_normal((64, 64)), 1.5) > 0.12)
        shapes.append(m)
    masks = eng.masks.clone()
    for i, m in enumerate(shapes):
        packed = np.packbits(m.reshape(64, 8, 8), axis=-1, bitorder='little').reshape(64, 8).view(np.uint32).reshape(64, 2)
        masks[0, i] = torch.from_numpy(packed.view(np.int32)).to(masks.device)
    eng.masks.copy_(masks)
    eng.counts[0] = len(shapes)
    got = eng.contours(1, cap=32, kept_only=False)[0]          # small cap: the checkerboard / noise overflow to the host path
    raw_n = eng.contour_n[0, :len(shapes)].cpu().numpy()
    assert (raw_n == -1).sum() >= 1 and (raw_n > 0).sum() >= 12
    for i, m in enumerate(shapes):
        assert np.array_equal(got[i], host.trace_outer_contour(m)), i
    got = eng.contours(1, cap=1024, kept_only=False)[0]
    assert (eng.contour_n[0, :len(shapes)].cpu().numpy() > 0).all()
    for i, m in enumerate(shapes):
        assert np.array_equal(got[i], host.trace_outer_contour(m)), i


def test_infer_cli_writes_overlays(hip_device, tmp_path):
    """tools/infer.py (BASELINE configs[0] plumbing): a folder of PNG tiles -> one overlay per image."""
    import subprocess
    import sys
    import torch
    from PIL import Image
    from nuhtc_amd import synth, weights
    tiles = synth.nuclei_tiles(4, 64, start=80)
    (tmp_path / 'imgs').mkdir()
    for i, t in enumerate(tiles):
        Image.fromarray(t).save(tmp_path / 'imgs' / f't{i}.png')
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools/infer.py'), str(tmp_path / 'imgs'), CFG, str(ck), '--output',
                                   str(tmp_path / 'o')], text=True)
    assert out.count('instances') == 4
    for i in range(4):
        im = np.array(Image.open(tmp_path / 'o' / f't{i}.png'))
        assert im.shape == (64, 64, 3)
    # a CPU device is an error, not a fallback
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools/infer.py'), str(tmp_path / 'imgs'), CFG, str(ck), '--device', 'cpu'],
                       capture_output=True, text=True)
    assert r.returncode != 0 and 'cpu' in (r.stderr + r.stdout).lower()


def test_real_checkpoint_parity_script(hip_device, tmp_path):
    """oracle/check_checkpoint.py (the one-command oracle-vs-HIP check for a supplied checkpoint) on a checkpoint file
    in the reference's format; synthetic weights stand in for models/pannuke.pth, which is not distributed."""
    import subprocess
    import sys
    import torch
    from nuhtc_amd import weights
    ck = tmp_path / 'ck.pth'
    sd = weights.bench_state_dict(0, obj_bias=0.0)
    torch.save(dict(meta=dict(CLASSES=('T', 'I', 'C', 'D', 'E')), state_dict={**sd, 'roi_head.kernel': torch.ones(1, 1, 5, 5),
                                                                                   'ema_backbone_norm0_weight': torch.zeros(96)}), ck)
    r = subprocess.run([sys.executable, '-m', 'oracle.check_checkpoint', '--checkpoint', str(ck), '--tiles', '3', '--size', '64'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and 'PARITY OK' in r.stdout, r.stderr[-1500:]


def _two_rank_env():
    """Two ranks on a one-GPU box: both on device 0, gloo instead of RCCL (RCCL refuses two ranks on one device)."""
    env = dict(os.environ)
    env.update(NUHTC_ONE_DEVICE='1', NUHTC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    return env


def _free_port():
    """A free TCP port on the loopback interface (as bench.self_launch picks one): concurrent test runs on one box, or a socket of an earlier
    run still in TIME_WAIT, must not make the rendezvous fail."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def test_infer_wsi_two_ranks_equal_one_rank_byte_for_byte(hip_device, tmp_path):
    """The multi-rank path end to end (north star: tiles sharded across ranks, one gather, merge on rank 0): tools/infer_wsi.py as
    two processes under torch.distributed.run writes the same files as one process, byte for byte -- per-tile GeoJSON, point
    GeoJSON and the merged slide -- and each rank cut only its own tiles (lazy sharded reads of the memory-mapped slide)."""
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, weights
    img = np.concatenate([np.concatenate([synth.nuclei_tile(40 + 3 * r + c, 128) for c in range(3)], 1) for r in range(2)], 0)   # 256 x 384
    src = tmp_path / 'slide.npy'
    np.save(src, img)
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    common = [os.path.join(ROOT, 'tools/infer_wsi.py'), str(src), CFG, str(ck), '--patch_size', '64', '--step_size', '48', '--batch_size', '4',
              '--merge', '--mode', 'qupath']
    subprocess.check_call([sys.executable] + common + ['--save_dir', str(tmp_path / 'one')])
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port())] + common + ['--save_dir', str(tmp_path / 'two')], env=_two_rank_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(out.stdout[-2000:])
    assert out.returncode == 0
    assert '48 tiles on 2 rank(s)' in out.stdout
    # `--gpus 2` without a launcher: the tool starts its own ranks ("add --gpus N")
    env = _two_rank_env()
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    out2 = subprocess.run([sys.executable] + common + ['--save_dir', str(tmp_path / 'self'), '--gpus', '2'], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    print(out2.stdout[-1500:])
    assert out2.returncode == 0 and '48 tiles on 2 rank(s)' in out2.stdout
    for f in ('slide.geojson', 'slide_point.geojson', 'slide_merged.geojson'):
        a = open(tmp_path / 'one/nuclei/slide' / f, 'rb').read()
        b = open(tmp_path / 'two/nuclei/slide' / f, 'rb').read()
        c = open(tmp_path / 'self/nuclei/slide' / f, 'rb').read()
        assert len(a) > 1000 and a == b == c, f


def test_eight_ranks_end_to_end_on_one_device(hip_device, tmp_path):
    """The 8-GPU job on the hardware a test box has: eight ranks under torch.distributed.run sharing the one MI355X (NUHTC_ONE_DEVICE=1,
    gloo for the exchange).  tools/infer_wsi.py --gpus 8 on a slide of SIX tiles (ranks 6 and 7 get none: empty shards travel through the
    engine pipeline, the record packing, the text writers and the gather) and on a slide of 48 tiles writes the documents of the one-rank
    run byte for byte, hears from every rank and ends (no rank left in a collective); bench.py --gpus 8 --batch 2 prints its one line with
    all eight ranks in the exchange.  No rate is claimed: the ranks share a GPU."""
    import json
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, weights
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0, obj_bias=0.0)), ck)
    env = _two_rank_env()
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env['OMP_NUM_THREADS'] = '2'
    small = np.concatenate([np.concatenate([synth.nuclei_tile(60 + 3 * r + c, 64) for c in range(3)], 1) for r in range(2)], 0)       # 128 x 192: 2 x 3 tiles of 64
    big = np.concatenate([np.concatenate([synth.nuclei_tile(40 + 3 * r + c, 128) for c in range(3)], 1) for r in range(2)], 0)        # 256 x 384: 6 x 8 tiles at stride 48
    for name, img, step, ntile, shares in (('six', small, 64, 6, [1, 1, 1, 1, 1, 1, 0, 0]), ('fortyeight', big, 48, 48, [6] * 8)):
        src = tmp_path / f'{name}.npy'
        np.save(src, img)
        common = [os.path.join(ROOT, 'tools/infer_wsi.py'), str(src), CFG, str(ck), '--patch_size', '64', '--step_size', str(step), '--batch_size', '4',
                  '--merge', '--mode', 'qupath']
        subprocess.check_call([sys.executable] + common + ['--save_dir', str(tmp_path / f'{name}_one')])
        out = subprocess.run([sys.executable] + common + ['--save_dir', str(tmp_path / f'{name}_eight'), '--gpus', '8'], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True, timeout=1500)
        print(out.stdout[-2500:])
        assert out.returncode == 0
        assert f'{ntile} tiles on 8 rank(s)' in out.stdout and f'records of ranks {list(range(8))}, tiles per rank {shares}' in out.stdout
        for f in (f'{name}.geojson', f'{name}_point.geojson', f'{name}_merged.geojson'):
            a = open(tmp_path / f'{name}_one/nuclei/{name}' / f, 'rb').read()
            b = open(tmp_path / f'{name}_eight/nuclei/{name}' / f, 'rb').read()
            assert a == b and len(a) > 2, f
        assert not [f for f in os.listdir(tmp_path / f'{name}_eight') if f.startswith('.host_phase_done')]       # rank 0 removed its marker
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--no-settle', '--no-fp32-pipe',
                          '--no-roi-load', '--batch', '2', '--in-flight', '2'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    print(out.stderr[-3000:])
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['exchange']['ranks_seen'] == list(range(8)) and d['exchange']['communicator_ranks'] == 8 and 'cpu_baseline' not in d
    assert abs(d['value'] - 8 * 2 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']


def test_bench_wsi_reads_an_svs_file_on_every_rank(hip_device):
    """tools/bench_wsi.py --svs with more than one rank: rank 0 writes the Aperio-layout file, the other rank waits for it on a marker file
    (before the process group exists) and every rank decodes only the tiles of its own shard through libtiff -- the way the ranks of
    tools/infer_wsi.py open a real slide.  Two ranks on the one device, gloo; the line reports both ranks' detections and the documents."""
    import json
    import subprocess
    import sys
    from nuhtc_amd import tiffslide
    if not tiffslide.available():
        pytest.skip('libtiff not found')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port()), os.path.join(ROOT, 'tools/bench_wsi.py'), '--grid', '6', '--svs', 'lzw', '--batch_size', '4',
                          '--depth', '2', '--workers', '2'], env=_two_rank_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    print(out.stderr[-2500:])
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['tiles'] == 36 and d['detections_after_tile_nms'] > 50 and 0 < d['detections_after_merge'] <= d['detections_after_tile_nms']
    assert 'libtiff' in d['tile_source'] and d['tiles_per_s_to_documents'] > 0 and set(d['document_bytes']) == {'slide.geojson', 'slide_point.geojson', 'slide_merged.geojson'}
    assert not [f for f in os.listdir('/tmp') if f.startswith('.host_phase_done.')]


def test_rccl_branch_of_the_exchange_on_one_gpu(hip_device):
    """The nccl (= RCCL) branch of `gather_blobs` on the hardware at hand: NUHTC_FORCE_COLLECTIVE=1 forms a communicator of one rank on
    the MI355X and both all_gathers run on DEVICE buffers; what comes back is byte-equal to the short-circuit path (the script of
    tests/test_host.py, which runs it over gloo on the CPU box)."""
    import subprocess
    import sys
    from test_host import _FORCED
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'NUHTC_FORCE_COLLECTIVE')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    out = subprocess.run([sys.executable, '-c', _FORCED, 'nccl'], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-500:], out.stderr[-1500:])
    assert out.returncode == 0 and 'FORCED OK nccl' in out.stdout


def test_bench_launches_its_own_ranks(hip_device):
    """`python bench.py --gpus 2` without a launcher: bench.py starts the two ranks itself (children under torch.distributed.run),
    relays rank 0's JSON line and exits 0; the line says n_gpus 2, both ranks came back through the gather, and the whole-job value
    counts both ranks' tiles.  (Both ranks share the one GPU of the test box, over gloo: the rate itself means nothing here.)"""
    import json
    import subprocess
    import sys
    env = _two_rank_env()
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-settle', '--no-fp32-pipe',
                          '--no-roi-load', '--batch', '4'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    print(out.stderr[-3000:])
    assert out.returncode == 0
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['exchange']['ranks_seen'] == [0, 1] and d['scaling'] == 'weak' and 'cpu_baseline' not in d
    assert abs(d['value'] - 2 * 4 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']


def test_batch_64_slide_equals_batch_8(hip_device, tmp_path):
    """BASELINE configs[4]'s shape (tools/infer_wsi.py, patch 256, step 192, batch_size 64): a pipeline of four max_batch=64 engines over
    a 12 x 12-tile slide gives, record for record, what batch_size 8 gives (tiles are independent: the batch size is a throughput
    knob, tools/infer_wsi.py:466-476), through the CLI with --merge."""
    import subprocess
    import sys
    import torch
    from nuhtc_amd import synth, weights
    G = 12
    full, _ = synth.nuclei_canvas(G)
    src = tmp_path / 'slide.npy'
    np.save(src, full)
    ck = tmp_path / 'w.pth'
    torch.save(dict(state_dict=weights.bench_state_dict(0)), ck)
    outs = {}
    for bs in (64, 8):
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools/infer_wsi.py'), str(src), CFG, str(ck), '--patch_size', '256', '--step_size', '192',
                               '--batch_size', str(bs), '--save_dir', str(tmp_path / f'b{bs}'), '--merge'])
        outs[bs] = {f: open(tmp_path / f'b{bs}/nuclei/slide' / f, 'rb').read() for f in ('slide.geojson', 'slide_merged.geojson')}
    assert len(outs[64]['slide.geojson']) > 100000
    assert outs[64] == outs[8]


@pytest.mark.gpu
def test_engine_stream_outlives_the_engine_and_gives_the_same_results(hip_device):
    """nuhtc_stream(): an engine run on the stream it owns gives the results of the default stream; the stream stays usable after
    the engine is closed (PyTorch's allocator touches the stream a block was allocated on when the block is freed) and the next
    engine of the device takes the pooled streams over."""
    import gc
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    sd = weights.bench_state_dict()
    tiles_np = synth.nuclei_tiles(2, 256, start=5)
    a = Engine(sd, device=0, max_batch=2, tile=(256, 256))
    tiles = a.to_device(tiles_np)
    a.infer_async(tiles, hip.CH_SWAP); a.check()
    ref = (a.counts.clone(), a.boxes.clone(), a.labels.clone())
    st = a.stream
    assert st.cuda_stream != 0 and st.cuda_stream != torch.cuda.current_stream().cuda_stream
    with torch.cuda.stream(st):
        t2 = a.to_device(tiles_np)                   # a block allocated on the engine's stream
        a.infer_async(t2, hip.CH_SWAP)
    st.synchronize(); a.check()
    assert torch.equal(a.counts, ref[0]) and torch.equal(a.boxes, ref[1]) and torch.equal(a.labels, ref[2])
    handle = st.cuda_stream
    a.close()
    del t2; gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()      # segfaulted when nuhtc_destroy destroyed the stream
    with torch.cuda.stream(st):
        assert float(torch.ones(8, device='cuda').sum()) == 8.0
    b = Engine(sd, device=0, max_batch=2, tile=(256, 256))
    assert b.stream.cuda_stream == handle                                           # the pooled triple
    with torch.cuda.stream(b.stream):
        b.infer_async(b.to_device(tiles_np), hip.CH_SWAP)
    b.stream.synchronize(); b.check()
    assert torch.equal(b.counts, ref[0]) and torch.equal(b.boxes, ref[1])
    b.close()


@pytest.mark.gpu
def test_throughput_schedule_is_bit_identical(hip_device):
    """nuhtc_config.schedule = NUHTC_SCHED_THROUGHPUT changes the block tiles of the Swin linears (256-row instead of 128-row) and keeps
    every kernel on the caller's stream: every output of the path must be the same bit for bit; an unknown value is refused."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine, HipError
    sd = weights.bench_state_dict()
    tiles_np = synth.nuclei_tiles(3, 256, start=11)
    a = Engine(sd, device=0, max_batch=3, tile=(256, 256))
    b = Engine(sd, device=0, max_batch=3, tile=(256, 256), schedule=hip.SCHED_THROUGHPUT)
    assert a.cfg.schedule == hip.SCHED_LATENCY and b.cfg.schedule == hip.SCHED_THROUGHPUT
    for e in (a, b):
        e.infer_async(e.to_device(tiles_np), hip.CH_SWAP); e.check()
    assert int(a.counts.sum()) > 20
    for name in ('c0', 'c3', 'x0', 'sem_feat', 'sem_pred', 'roi_counts'):
        assert torch.equal(a.buffer(name), b.buffer(name)), name
    n = int(a.buffer('roi_total').item())                    # (rows past the total are workspace)
    assert n == int(b.buffer('roi_total').item()) and torch.equal(a.buffer('rois')[:n], b.buffer('rois')[:n])
    for f in ('counts', 'boxes', 'labels', 'masks', 'keep'):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    # RoIs of every size class through the fixed-load entry point: the single-stream order of the RoI kernels gives the same features
    rng = np.random.default_rng(5)
    n = 300
    wh = np.concatenate([rng.uniform(8, 44, (n // 3, 2)), rng.uniform(44, 112, (n // 3, 2)), rng.uniform(112, 300, (n // 3, 2))]).astype(np.float32)
    ctr = rng.uniform(0, 512, (3, n, 2)).astype(np.float32)
    rois = np.clip(np.concatenate([ctr - wh[None] / 2, ctr + wh[None] / 2], -1), 0, 512).astype(np.float32)
    rois[:, :, 2:] = np.maximum(rois[:, :, 2:], rois[:, :, :2] + 4)
    for e in (a, b):
        t = e.to_device(tiles_np)
        e.infer_fixed_load_async(t, torch.from_numpy(rois).to(t.device), 30, hip.CH_SWAP); e.check()
    ca = a.buffer('roi_fallback_count').cpu().numpy()
    assert ca[0] > 50 and ca[1] > 50, ca
    assert torch.equal(a.buffer('bbox_feats')[:3 * n], b.buffer('bbox_feats')[:3 * n])
    assert torch.equal(a.boxes, b.boxes) and torch.equal(a.masks, b.masks)
    a.close(); b.close()
    with pytest.raises(HipError):
        Engine(sd, device=0, max_batch=1, tile=(256, 256), schedule=7)


def test_build_then_smoke_in_one_process(hip_device):
    """The driver's two entry points in ONE process, build() first: build() loads libnuhtc_hip.so to check its exports, and if that happened
    before torch was imported the process held two HIP runtimes (torch brings its own copy) and the engine found no device.  hip.load()
    imports torch first."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    out = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.build(); g.smoke()'], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'smoke ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
