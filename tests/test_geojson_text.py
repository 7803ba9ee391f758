"""The library's GeoJSON text writers (nuhtc_write_ring_features / nuhtc_write_point_features / nuhtc_join_features, csrc/geojson.hip) against
what they replace: json.dumps of the feature dicts tools/infer_wsi.py:550-585 of the reference builds (nuhtc_amd.contours.feature /
point_feature, themselves pinned to the reference's writer in tests/test_contours.py).  Host code only: runs without a GPU."""
import json

import numpy as np

from nuhtc_amd import contours

CLASSES = ('T', 'I', 'C', 'D', 'E')


def _records(n, seed):
    rng = np.random.default_rng(seed)
    ring_n = rng.integers(1, 70, n)
    verts = rng.integers(-5, 200000, (int(ring_n.sum()), 2)).astype(np.int32)
    labels = rng.integers(0, 5, n).astype(np.int32)
    scores = rng.random(n).astype(np.float32).astype(np.float64)          # what the records carry: float32 scores widened
    boxes = (rng.random((n, 4)).astype(np.float32) * 256).astype(np.float64) + rng.integers(0, 100000, (n, 1))
    return ring_n, verts, labels, scores, boxes


def _dicts(ring_n, verts, labels, scores):
    off = np.concatenate([[0], np.cumsum(ring_n)])
    return [contours.feature(verts[off[i]:off[i + 1]], labels[i], scores[i], CLASSES) for i in range(len(ring_n))]


def test_ring_and_point_text_equal_json_dumps():
    ring_n, verts, labels, scores, boxes = _records(3000, 0)
    ring_n[7] = 0                                              # an empty ring: "coordinates": [[]]
    verts = verts[:int(ring_n.sum())]
    verts[:2] = [[0, -1], [-2147483648, 2147483647]]
    scores[:7] = [1.0, 0.5, 1e-5, 0.35, 0.0001, 1e16, 0.30000000000000004]
    boxes[0] = [10, 20, 30, 40]                                # integral centre -> "20.0"
    feats = _dicts(ring_n, verts, labels, scores)
    for threads in (1, 0):
        text, start = contours.ring_features_text(verts, ring_n, labels, scores, CLASSES, threads=threads)
        assert '[' + text.tobytes().decode() + ']' == json.dumps(feats)
        for i in (0, 7, 1234, len(feats) - 1):
            assert text[start[i]:start[i + 1] - 2].tobytes().decode() == json.dumps(feats[i])
    pt = contours.point_features_text(boxes, labels, scores, CLASSES)
    assert '[' + pt.tobytes().decode() + ']' == json.dumps([contours.point_feature(boxes[i], labels[i], scores[i], CLASSES) for i in range(len(labels))])


def test_float_text_is_python_repr():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(20000), rng.random(20000).astype(np.float32).astype(np.float64), np.exp(rng.uniform(-60, 60, 40000)),
                        rng.integers(0, 10 ** 7, 20000) / 2.0, -rng.random(1000), [0.0, -0.0, 1e-4, 9.999e-5, 1e16, 9999999999999998.0, 5e-324, 8.98e307]])
    t = contours.point_features_text(np.stack([x, x, x, x], 1), np.zeros(len(x), np.int32), x, CLASSES)
    assert '[' + t.tobytes().decode() + ']' == json.dumps([contours.point_feature([v, v, v, v], 0, v, CLASSES) for v in x.tolist()])


def test_rank_texts_concatenate_and_join_like_one_list():
    """Three ranks (the middle one without records): concatenation = the dump of all records; the merged document = the dump of the kept ones."""
    parts = [_records(500, 2), None, _records(300, 3)]
    texts, starts, feats = [], [], []
    for p in parts:
        if p is None:
            t, s = contours.ring_features_text(np.zeros((0, 2), np.int32), np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0), CLASSES)
        else:
            t, s = contours.ring_features_text(p[1], p[0], p[2], p[3], CLASSES)
            feats += _dicts(p[0], p[1], p[2], p[3])
        texts.append(t)
        starts.append(s)
    body, start = contours.concat_feature_texts(texts, starts)
    assert '[' + body.tobytes().decode() + ']' == json.dumps(feats)
    kept = np.array([0, 3, 499, 500, 650, 799])
    assert '[' + contours.join_features_text(body, start, kept).tobytes().decode() + ']' == json.dumps([feats[i] for i in kept])
    assert len(contours.join_features_text(body, start, np.zeros(0, np.int64))) == 0
    empty, es = contours.concat_feature_texts([texts[1]], [starts[1]])
    assert len(empty) == 0 and es.tolist() == [2]


def test_merge_features_leaves_its_input_alone():
    ring_n, verts, labels, scores, _ = _records(40, 4)
    sq = lambda x, y, s: np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s], [x, y]])
    feats = [contours.feature(sq(10 * i, 0, 8), labels[i], scores[i], CLASSES) for i in range(10)]
    before = json.dumps(feats)
    out = contours.merge_features(feats, 0.05)
    assert json.dumps(feats) == before and all('nuclei_id' in f['properties'] for f in out)
    assert list(out[0]['properties'])[-1] == 'nuclei_id'


def test_filled_rings_are_the_traced_components_with_their_holes_filled():
    """nuhtc_fill_rings (what lets tools/nuclei_merge.py hand a written GeoJSON to the GPU merge): the pixels inside or on the traced outer
    border of an 8-connected component are that component with its holes filled -- scipy's binary_fill_holes (4-connected background, like
    OpenCV's) of the component, on random fragmented masks, crops and areas in the packed layout; anything that is not a traced ring is refused."""
    from scipy import ndimage as ndi
    from nuhtc_amd import wsi
    rng = np.random.default_rng(5)
    rings, want = [], []
    for _ in range(200):
        h, w = rng.integers(3, 40, 2)
        m = rng.random((h, w)) < rng.uniform(0.3, 0.8)
        lab, k = ndi.label(m, structure=np.ones((3, 3)))
        if k == 0:
            continue
        comp = lab == lab[np.nonzero(m)][0]
        ring = contours.trace_outer_contour(comp)
        shift = rng.integers(-50, 5000, 2)
        rings.append(np.concatenate([ring, ring[:1]]) + shift)              # closed, as the files hold them
        f = ndi.binary_fill_holes(comp)
        ys, xs = np.nonzero(f)
        want.append((f[ys.min():ys.max() + 1, xs.min():xs.max() + 1], xs.min() + shift[0], ys.min() + shift[1]))
    boxes, areas, bits, off = contours.fill_rings(rings)
    got = wsi.PackedMasks(boxes, areas, bits, off)
    assert len(got) == len(want) > 150
    for (m, x0, y0), (wm, wx, wy), a in zip(got, want, areas):
        assert (x0, y0) == (wx, wy) and np.array_equal(m, wm) and a == wm.sum()
    assert contours.fill_rings([np.array([[0, 0], [5, 1], [3, 3]])]) is None                # an edge off the chain directions
    assert contours.fill_rings([np.array([[0.5, 0], [4, 0], [4, 4]])]) is None              # not on pixel centres
    one = contours.fill_rings([np.array([[7, 9], [7, 9]])])                                 # a one-pixel component
    assert one[0].tolist() == [[7, 9, 8, 10]] and one[1].tolist() == [1]
