"""nuhtc_amd.evaluation against golden vectors produced by the reference's own metric code
(oracle/ref_harness/make_eval_golden.py; tools/analysis_tools/pannuke/utils.py, nuhtc/utils/stats_utils.py).
Tolerance: counts exact, ratios 1e-12 relative (same float64 arithmetic, different summation order in the matrix product)."""
import os

import numpy as np
import pytest

from nuhtc_amd import evaluation as E

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def ml():
    return np.load(os.path.join(G, 'eval_masklist.npz'))


@pytest.fixture(scope='module')
def pk():
    return np.load(os.path.join(G, 'eval_pannuke.npz'))


def test_pairwise_and_masklist_stats(ml):
    for i in range(int(ml['n_img'])):
        t, p = ml[f'true{i}'], ml[f'pred{i}']
        inter, union = E.pairwise_inter_union(t, p)
        assert np.array_equal(inter, ml[f'inter{i}']) and np.array_equal(union, ml[f'union{i}'])
        s = E.stat_calc(t, p)
        assert s['aji'] == pytest.approx(float(ml[f'aji{i}']), rel=1e-12)
        assert s['aji_plus'] == pytest.approx(float(ml[f'aji_plus{i}']), rel=1e-12)
        assert [s['dq'], s['sq'], s['pq']] == pytest.approx(list(ml[f'pq{i}']), rel=1e-12)
        assert [s['tp'], s['fp'], s['fn']] == list(ml[f'pq_counts{i}'])
        assert s['dice'] == pytest.approx(float(ml[f'dice{i}']), rel=1e-12)
        assert s['iou'] == pytest.approx(float(ml[f'pq{i}'][1]) * (s['tp'] + 1e-6), rel=1e-12)
        # Munkres pairing (no pairing passed in), match_iou below 0.5
        assert E.get_fast_pq(t, p, match_iou=0.3)[0] == pytest.approx(list(ml[f'pq_munkres{i}']), rel=1e-12)


def test_stat_calc_empty_sides():
    m = np.zeros((2, 8, 8), np.uint8)
    m[0, :3, :3] = 1
    m[1, 5:, 5:] = 1
    assert E.stat_calc([], []) is None
    assert E.stat_calc([], m)['fp'] == 2 and E.stat_calc([], m)['pq'] == 0
    assert E.stat_calc(m, [])['fn'] == 2
    s = E.stat_calc(m, m)
    assert s['pq'] == pytest.approx(1.0, abs=1e-6) and s['aji'] == pytest.approx(1.0) and s['dice'] == pytest.approx(1.0)


def test_pannuke_protocol(pk):
    true, pred, types = pk['true'], pk['pred'], list(pk['types'])
    assert np.array_equal(E.binarize(true[0, :, :, :5]), pk['binarize0'])
    assert np.array_equal(E.remap_label(E.binarize(pred[0, :, :, :5])), pk['remap0'])
    res = E.pannuke_stats(true, pred, types, num_classes=5)
    np.testing.assert_allclose(res['class_pq'], pk['class_pq'], rtol=1e-12, equal_nan=True)
    for n, m, b in zip(pk['tissue_names'], pk['tissue_mpq'], pk['tissue_bpq']):
        np.testing.assert_allclose(res['tissue_mpq'][str(n)], m, rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(res['tissue_bpq'][str(n)], b, rtol=1e-12, equal_nan=True)
    # per (image, class) pairing counts
    for i, c, tp, fp, fn in pk['pairs']:
        r = E.get_fast_pq_map(E.remap_label(true[i, :, :, c]), E.remap_label(pred[i, :, :, c]))
        assert [len(r[1][0]), len(r[1][3]), len(r[1][2])] == [tp, fp, fn]
    assert E.get_fast_pq_map(pk['t0'], pk['p0'], match_iou=0.3)[0] == pytest.approx(list(pk['pq_munkres']), rel=1e-12)


def test_convert_format_and_multiclass():
    rng = np.random.default_rng(0)
    H = W = 32
    masks = np.zeros((4, H, W), bool)
    masks[0, 2:8, 2:8] = True
    masks[1, 6:12, 6:12] = True       # overlaps instance 0
    masks[2, 20:26, 3:9] = True
    masks[3, 20:30, 20:30] = True
    labels = np.array([0, 2, 0, 1])
    pn = E.convert_format(masks, labels, H, W, 3, 'pannuke')
    assert pn.shape == (H, W, 4)
    assert pn[3, 3, 0] == 1 and pn[22, 5, 0] == 2 and pn[7, 7, 2] == 1 and pn[25, 25, 1] == 1
    assert pn[7, 7, 0] == 1                                # per-class channels keep both overlapping instances
    assert pn[0, 0, 3] == 1 and pn[7, 7, 3] == 0           # background channel
    co = E.convert_format(masks, labels, H, W, 3, 'conic')
    assert co[7, 7, 0] == 2 and co[7, 7, 1] == 3           # later instance wins, class + 1
    cs = E.convert_format(masks, labels, H, W, 3, 'consep')
    assert cs['inst_centroid'][0].tolist() == [5.0, 5.0] and len(cs['inst_uid']) == 3
    assert E.convert_format(np.zeros((0, H, W)), [], H, W, 3, 'pannuke').sum() == 0
    # multi-class table + aggregation: perfect prediction -> PQ 1 for present classes
    info = E.multi_stat_calc(masks, masks, labels, labels, 4)
    assert info[0][:3] == [2, 0, 0] and np.isnan(info[3][0])
    agg = E.aggregate_mpq([info, info])
    assert agg['multi_pq+_0'] == pytest.approx(1.0, abs=1e-5) and agg['multi_pq+_3'] == 0.0
    cm = E.update_confusion_matrix(np.zeros((4, 4)), masks, masks[:3], labels, np.array([0, 1, 0]))
    assert cm[0, 0] == 2 and cm[2, 1] == 1 and cm[1, 3] == 1   # class-2 instance predicted as 1; instance 3 missed
    kept, sel = E.mask_post_process(np.concatenate([masks, masks[:1, :, :] & masks[:1, :, :]]), 4)
    assert sel.sum() == 3                                   # the duplicated instance removes both copies of itself
