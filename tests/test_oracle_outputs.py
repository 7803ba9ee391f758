"""The product's output writers (nuhtc_amd.contours.feature / point_feature, nuhtc_amd.outputs, nuhtc_amd.cocomask) against
oracle/writers.py -- the reference's inline writer code of tools/infer_wsi.py:541-693 and tools/infer_patch.py:271-290 restated
statement by statement -- and oracle/rle.py (cocoapi maskApi.c restated as scalar loops), on the same random detections: JSON
documents equal as Python objects (key for key, value for value and type for type after a json round trip), SQLite rows equal
column for column.  CPU only."""
import json
import sqlite3

import numpy as np

from nuhtc_amd import cocomask, contours, outputs
from oracle import rle as R
from oracle import writers as W

CLASSES = ('T', 'I', 'C', 'D', 'E')


def _blob(rng, P):
    m = np.zeros((P, P), bool)
    y, x = rng.integers(1, P - 14, 2)
    h, w = rng.integers(3, 12, 2)
    yy, xx = np.mgrid[0:P, 0:P]
    m |= ((yy - y - h / 2) / (h / 2 + 0.3)) ** 2 + ((xx - x - w / 2) / (w / 2 + 0.3)) ** 2 <= 1
    if rng.uniform() < 0.3:
        m[y + h // 2, x:x + w] = False          # split into fragments: the first contour of the list is what gets written
    return m


def _tile(rng, P=64, k=9):
    masks = [_blob(rng, P) for _ in range(k)]
    masks = [m for m in masks if m.any()]
    k = len(masks)
    labels = rng.integers(0, 5, k)
    scores = rng.uniform(0.35, 1.0, k)
    boxes = np.zeros((k, 4))
    for i, m in enumerate(masks):
        ys, xs = np.nonzero(m)
        boxes[i] = [xs.min() - 0.3, ys.min() - 0.2, xs.max() + 1.4, ys.max() + 1.1]
    return masks, labels, scores, boxes


def test_rle_codec_against_the_scalar_restatement():
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, w = rng.integers(1, 40, 2)
        m = rng.uniform(size=(h, w)) < rng.uniform()
        a, b = cocomask.encode(m), R.encode(m.tolist())
        assert a == b
        cnts = R.rle_fr_string(a['counts'])
        assert cocomask.string_to_counts(a['counts']) == cnts and cocomask.area(a) == R.rle_area(cnts) == int(m.sum())
        assert cocomask.to_bbox(a) == R.rle_to_bbox(h, w, cnts)
    # hand-worked strings of maskApi.c rleToString
    assert R.rle_to_string([1, 3]) == '13' and R.rle_to_string([32]) == 'P1' and R.rle_to_string([16]) == '`0'
    assert R.rle_to_string([5, 9, 7, 3]) == '597' + chr(48 + 0x1a) and R.rle_fr_string('597' + chr(48 + 0x1a)) == [5, 9, 7, 3]
    # pairwise IoU: the run-walking routine against the product's dense evaluation
    ms = [rng.uniform(size=(17, 23)) < 0.4 for _ in range(6)]
    rl = [cocomask.encode(m) for m in ms]
    dense = cocomask.iou(rl, rl)
    for i in range(6):
        for j in range(6):
            want = R.rle_iou(R.rle_fr_string(rl[i]['counts']), R.rle_fr_string(rl[j]['counts']))
            assert abs(dense[i, j] - want) < 1e-12


def test_infer_wsi_documents_field_by_field(tmp_path):
    rng = np.random.default_rng(1)
    P = 64
    geo, pts, dsa, imgs, annts, rows = [], [], [], [], [], []        # product
    o_geo, o_pts, o_dsa, o_imgs, o_annts, o_rows = [], [], [], [], [], []
    sql = outputs.SqlContourWriter(str(tmp_path / 'p.db'))
    nid = 0
    for annidx, origin in enumerate([(0, 0), (192, 0), (1000, 2304)]):
        masks, labels, scores, boxes = _tile(rng, P)
        rings = [contours.mask_to_ring(m, origin=origin) for m in masks]                          # mask2inst + coord
        keep = [i for i, r in enumerate(rings) if len(r) >= 3]                                    # :536 (`len(con) >= 3` on the closed contour)
        org4 = np.array([origin[0], origin[1], origin[0], origin[1]], np.float64)
        per_tile = []
        for e, i in enumerate(keep):
            geo.append(contours.feature(rings[i], labels[i], scores[i], CLASSES))
            pts.append(contours.point_feature(boxes[i] + org4, labels[i], scores[i], CLASSES))
            dsa.append(outputs.dsa_element(rings[i], labels[i], CLASSES))
            an = outputs.coco_annotation(masks[i], labels[i], annidx, nid + e)
            annts.append(an)
            sql.add(annidx, e, rings[i], labels[i], scores[i], CLASSES)
            per_tile.append(int(labels[i]))
        imgs.append(outputs.coco_tile_image(annidx, P, P, per_tile, CLASSES))
        # the reference's statements on the same detections
        d = W.tile_documents([rings[i].reshape(1, -1, 2) for i in keep], boxes[keep] + org4, labels[keep], scores[keep], annidx, P, P,
                             rle_inst=[R.encode(masks[i].tolist()) for i in keep], nuclei_id=nid)
        nid += len(keep)
        o_geo += d['geojson']; o_pts += d['pointjson']; o_dsa += d['dsajson']; o_imgs.append(d['img_dict']); o_annts += d['annts']; o_rows += d['sql_rows']
    sql.close()
    rt = lambda x: json.loads(json.dumps(x))
    assert len(geo) > 15
    assert rt(geo) == rt(o_geo) and rt(pts) == rt(o_pts)
    assert rt(outputs.dsa_document(dsa)) == rt(W.dsa_file(o_dsa))
    assert rt({'images': imgs, 'annotations': annts, 'categories': outputs.coco_categories(CLASSES)}) == rt(W.coco_file(o_imgs, o_annts))
    # key order of a feature as the reference writes it (QuPath reads by name; kept anyway)
    assert list(geo[0]) == list(o_geo[0]) and list(geo[0]['properties']) == list(o_geo[0]['properties']) and list(dsa[0]) == list(o_dsa[0])
    c = sqlite3.connect(str(tmp_path / 'p.db'))
    got = c.execute('SELECT annidx, elementidx, type, "group", score, color, xmin, ymin, xmax, ymax, bbox_area, coords_x, coords_y, keep FROM contour ORDER BY id').fetchall()
    assert got == o_rows
    n_rtree = c.execute('SELECT COUNT(*) FROM rtree').fetchone()[0]
    assert n_rtree == len(o_rows)
    r0 = c.execute('SELECT id, xmin, xmax, ymin, ymax FROM rtree ORDER BY id LIMIT 1').fetchone()
    assert r0[0] == 1 and (r0[1], r0[2], r0[3], r0[4]) == (float(o_rows[0][6]), float(o_rows[0][8]), float(o_rows[0][7]), float(o_rows[0][9]))


def test_infer_patch_annotations_field_by_field():
    rng = np.random.default_rng(2)
    masks, labels, scores, _ = _tile(rng, 48, 7)
    for k, (m, l, s) in enumerate(zip(masks, labels, scores)):
        got = outputs.coco_annotation(m, l, image_id=3, ann_id=10 + k, score=s)
        want = W.infer_patch_annotation(m.tolist(), l, s, 3, 10 + k)
        assert json.loads(json.dumps(got)) == json.loads(json.dumps(want))
