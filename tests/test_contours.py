"""CPU tests of the contour / polygon / GeoJSON host code."""
import json
import os
import subprocess
import sys

import numpy as np

from nuhtc_amd import contours as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _disk(r, size):
    yy, xx = np.mgrid[0:size, 0:size]
    return (yy - size // 2) ** 2 + (xx - size // 2) ** 2 <= r * r


def test_contour_of_rectangle_is_its_four_corners():
    m = np.zeros((10, 12), bool)
    m[2:7, 3:9] = True
    c = C.trace_outer_contour(m)
    assert sorted(map(tuple, c)) == [(3, 2), (3, 6), (8, 2), (8, 6)]
    ring = C.mask_to_ring(m, origin=(100, 200))
    assert np.array_equal(ring[0], ring[-1]) and ring[:, 0].min() == 103 and ring[:, 1].max() == 206


def test_contour_edge_cases():
    assert len(C.trace_outer_contour(np.zeros((5, 5), bool))) == 0
    one = np.zeros((5, 5), bool); one[2, 3] = True
    assert C.trace_outer_contour(one).tolist() == [[3, 2]]
    line = np.zeros((5, 8), bool); line[1, 2:7] = True
    assert sorted(map(tuple, C.trace_outer_contour(line))) == [(2, 1), (6, 1)]
    full = np.ones((4, 4), bool)
    assert sorted(map(tuple, C.trace_outer_contour(full))) == [(0, 0), (0, 3), (3, 0), (3, 3)]


def test_contour_raster_roundtrip_matches_mask():
    # polygon through the border-pixel centres, rasterised at pixel centres, reproduces the blob up to its 1-px rim
    for r in (4, 9, 15):
        m = _disk(r, 48)
        ring = C.mask_to_ring(m)
        ras = C.rasterize_polygon(ring.astype(float) + 0.5, 0, 0, 48, 48)   # vertices are pixel indices -> centres
        inter = np.logical_and(ras, m).sum()
        assert inter == ras.sum()                      # never outside the mask
        assert inter / m.sum() > 1 - 4.0 / r           # misses at most the outer half-pixel rim
    # concave blob: two overlapping disks
    m = np.zeros((40, 60), bool)
    m[:, :40] |= _disk(12, 40); m[:, 20:] |= _disk(12, 40)
    ring = C.mask_to_ring(m)
    ras = C.rasterize_polygon(ring.astype(float) + 0.5, 0, 0, 60, 40)
    assert np.logical_and(ras, ~m).sum() == 0 and ras.sum() / m.sum() > 0.8


def test_polygon_iou_known_answers():
    sq = lambda x, y, s: np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s], [x, y]], float)
    assert abs(C.polygon_iou(sq(0, 0, 10), sq(0, 0, 10)) - 1.0) < 1e-9
    assert C.polygon_iou(sq(0, 0, 10), sq(20, 20, 10)) == 0.0
    assert abs(C.polygon_iou(sq(0, 0, 10), sq(5, 0, 10)) - 50 / 150) < 1e-9


def test_merge_features_strategies_and_cli(tmp_path):
    names = ['T', 'I', 'C', 'D', 'E']
    sq = lambda x, y, s: np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s], [x, y]], float)
    feats = [C.feature(sq(0, 0, 10), 0, 0.9, names), C.feature(sq(2, 0, 14), 1, 0.8, names),     # overlap, second is larger
             C.feature(sq(100, 100, 10), 2, 0.95, names), C.feature(sq(10, 0, 5), 3, 0.5, names)]   # touches only -> kept
    prob = C.merge_features(feats, 0.05, 'probability')
    assert [f['properties']['label'] for f in prob] == [2, 0] + [3]
    assert [f['properties']['nuclei_id'] for f in prob] == [0, 1, 3]
    area = C.merge_features(feats, 0.05, 'area')
    assert sorted(f['properties']['label'] for f in area) == [1, 2, 3]
    p = tmp_path / 's.geojson'
    p.write_text(json.dumps(feats))
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools/nuclei_merge.py'), '--geojson', str(p), '--overlap_threshold', '0.05'])
    out = json.loads((tmp_path / 's_merged.geojson').read_text())
    assert len(out) == 3 and all('nuclei_id' in f['properties'] for f in out)
    pf = C.point_feature([0, 0, 10, 20], 1, 0.5, names)
    assert pf['geometry']['coordinates'] == [5.0, 10.0]
