"""Pins the oracle (oracle/model.py) against golden vectors produced by the reference's own Python files
(oracle/ref_harness/make_golden.py, run in the build container).  CPU-only."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import model as O

CASES = ['small_b2', 'small_wsi_b3', 'full_b1', 'five_b2', 'pad_b2']   # pad_b2: 72 x 90 image, Pad(size_divisor=32) acts (img_shape != pad_shape)


@pytest.fixture(scope='module', params=CASES)
def run(request):
    g = G.load(request.param)
    sd = G.seeded_sd(g)
    tiles = g['tiles']
    img = O.preprocess(tiles, int(g['channel_mode']))
    res, it = O.Oracle(sd).forward_tensor(img, tiles.shape[1:3], keep=True, img_hw=(2 * tiles.shape[1], 2 * tiles.shape[2]))
    _, toks = O.backbone(sd, img, return_tokens=True)
    return g, res, it, toks


def test_backbone_tokens(run):
    g, res, it, toks = run
    G.check_sub(g, 'embed', toks['embed'], 1e-5, 1e-5)
    for s in range(4):
        for b in range(O.DEPTHS[s]):
            G.check_sub(g, f's{s}b{b}', toks[f's{s}b{b}'], 1e-5, 1e-5)


def test_dense_stages(run):
    g, res, it, _ = run
    for i in range(4):
        G.check_sub(g, f'c{i}', it['c'][i], 1e-5, 1e-5)
        G.check_sub(g, f'x{i}', it['x'][i], 1e-5, 1e-5)
        G.check_sub(g, f'rpn_cls{i}', it['rpn_cls'][i], 1e-5, 1e-5)
        G.check_sub(g, f'rpn_reg{i}', it['rpn_reg'][i], 1e-5, 1e-5)
    G.check_sub(g, 'sem_pred', it['sem_pred'], 1e-5, 1e-5)
    G.check_sub(g, 'sem_feat', it['sem_feat'], 1e-5, 1e-5)


def test_proposals(run):
    g, res, it, _ = run
    for i in range(len(res)):
        # exact score ties (saturated sigmoid) are ordered by torch's unstable sort in the reference and by
        # "lower index first" in the oracle/HIP engine: compare with tied rows put in a canonical order
        np.testing.assert_allclose(G.canon_rows(it['rpn'][i].numpy()), G.canon_rows(g[f'rpn_props{i}']), rtol=0, atol=1e-4)
        np.testing.assert_array_equal(it['ws'][i].numpy(), g[f'ws{i}'])


def test_cascade(run):
    g, res, it, _ = run
    def canon(cls, reg):   # row order follows proposal order, which may swap exact score ties (see test_proposals)
        a = np.concatenate([cls, reg], 1)
        return a[np.lexsort((np.round(a[:, 2], 3), np.round(a[:, 1], 3), np.round(a[:, 0], 3)))]
    for k in range(3):
        np.testing.assert_allclose(canon(it['stage_cls'][k].numpy(), it['stage_reg'][k].numpy()),
                                   canon(g[f'cls{k}'], g[f'reg{k}']), rtol=0, atol=2e-5)


def test_detections_and_masks(run):
    g, res, it, _ = run
    G.check_prob_vs_logits(g, 'mask_logits', it['mask_prob'], atol=1e-6)
    for i, (br, sr) in enumerate(res):
        det = np.concatenate(br, 0)
        lab = np.concatenate([np.full(len(b), c, np.int32) for c, b in enumerate(br)])
        np.testing.assert_allclose(det, g[f'det{i}'], rtol=0, atol=1e-4)
        np.testing.assert_array_equal(lab, g[f'lab{i}'])
        ms = [m for cl in sr for m in cl]
        gm = np.unpackbits(g[f'masks{i}'], axis=-1).astype(bool)[..., :g['tiles'].shape[2]]
        assert len(ms) == len(gm)
        if len(ms):
            np.testing.assert_array_equal(np.stack(ms), gm)   # bit-exact masks


def test_result_format(run):
    g, res, it, _ = run
    for br, sr in res:
        assert len(br) == 5 and len(sr) == 5
        for b, s in zip(br, sr):
            assert b.dtype == np.float32 and b.shape[1] == 5 and len(s) == len(b)
            for m in s:
                assert m.dtype == bool and m.shape == tuple(g['tiles'].shape[1:3])
