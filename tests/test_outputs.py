"""CPU tests of the COCO RLE restatement (cocoapi maskApi.c; parity unpinned: pycocotools is absent) and the DSA / COCO /
SQL writers (tools/infer_wsi.py:587-693)."""
import sqlite3

import numpy as np
import pytest

from nuhtc_amd import cocomask as cm
from nuhtc_amd import outputs


def test_rle_string_hand_worked_vectors():
    # maskApi.c rleToString worked by hand: value < 16 -> one char; 32 -> low 5 bits 0 + continuation, then 1
    assert cm.counts_to_string([1, 3]) == '13'
    assert cm.counts_to_string([0, 4]) == '04'
    assert cm.counts_to_string([32]) == 'P1'
    assert cm.counts_to_string([16]) == '`0'          # bit 4 set -> needs a continuation (0x10 | 0x20, +48) so the sign is not misread
    # fourth run is stored as a difference from the second: [5, 2, 7, 2] -> '5', '2', '7', 0
    assert cm.counts_to_string([5, 2, 7, 2]) == '5270'
    # negative difference: [5, 9, 7, 3] -> 3 - 9 = -6 -> two's complement low bits 11010 with sign bit set, no continuation
    assert cm.counts_to_string([5, 9, 7, 3]) == '597' + chr(48 + 0x1a)
    for counts in ([1, 3], [32], [16], [5, 2, 7, 2], [5, 9, 7, 3], [0, 70000, 3, 1, 69000, 2]):
        assert cm.string_to_counts(cm.counts_to_string(counts)) == counts


def test_rle_round_trip_area_bbox_iou():
    rng = np.random.default_rng(0)
    for _ in range(100):
        h, w = rng.integers(1, 48, 2)
        m = (rng.uniform(size=(h, w)) < rng.uniform()).astype(np.uint8)
        r = cm.encode(m)
        assert r['size'] == [h, w] and np.array_equal(cm.decode(r), m) and cm.area(r) == m.sum()
        bb = cm.to_bbox(r)
        if m.sum():
            ys, xs = np.nonzero(m)
            assert bb == [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1]
        else:
            assert bb == [0, 0, 0, 0]
    a = np.zeros((8, 8), np.uint8); a[:4, :4] = 1
    b = np.zeros((8, 8), np.uint8); b[2:6, 2:6] = 1
    iou = cm.iou([cm.encode(a), cm.encode(b)], [cm.encode(a)])
    assert iou[0, 0] == 1.0 and iou[1, 0] == pytest.approx(4 / 28)
    # column-major order: a single pixel at (y=1, x=0) of a 3x2 mask is the second element
    assert cm.string_to_counts(cm.encode(np.array([[0, 0], [1, 0], [0, 0]]))['counts']) == [1, 1, 4]


def test_writers(tmp_path):
    classes = ('T', 'I', 'C', 'D', 'E')
    ring = np.array([[10, 10], [14, 10], [14, 13], [10, 13], [10, 10]])
    el = outputs.dsa_element(ring, 1, classes)
    assert el['points'][1] == [14, 10, 0] and el['closed'] and el['label'] == {'value': 'I'} and el['lineColor'] == 'rgb(0, 255, 0)'
    assert outputs.dsa_document([el])['name'] == 'NuHTC'
    m = np.zeros((16, 16), bool); m[3:7, 5:11] = True
    an = outputs.coco_annotation(m, 2, image_id=7, ann_id=3, score=0.9)
    assert an['bbox'] == [5, 3, 6, 4] and an['area'] == 24 and an['category_id'] == 2 and an['score'] == pytest.approx(0.9)
    assert np.array_equal(cm.decode(an['segmentation']), m)
    im = outputs.coco_tile_image(4, 16, 16, [2, 2, 0], classes)
    assert im == {'file_name': '4.png', 'height': 16, 'width': 16, 'id': 4, 'n_objects': 3, 'type': 'C'}
    db = str(tmp_path / 'c.db')
    w = outputs.SqlContourWriter(db)
    w.add(0, 0, ring, 1, 0.8, classes)
    w.add(0, 1, ring + 100, 0, 0.7, classes)
    w.close()
    c = sqlite3.connect(db)
    rows = c.execute('SELECT annidx, elementidx, "group", xmin, ymin, xmax, ymax, bbox_area, coords_x, keep FROM contour ORDER BY id').fetchall()
    assert rows[0] == (0, 0, 'I', 10, 10, 14, 13, 12, '10,14,14,10,10', 1) and rows[1][2] == 'T'
    hit = c.execute('SELECT id FROM rtree WHERE xmin <= 12 AND xmax >= 12 AND ymin <= 11 AND ymax >= 11').fetchall()
    assert hit == [(1,)]
