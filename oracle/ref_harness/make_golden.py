"""Generate golden vectors by running the REFERENCE's own Python files (build container only).

    python -m oracle.ref_harness.make_golden            # writes tests/golden/*.npz

The reference is imported from /root/reference under the mmcv stub (mmcv_stub.py); weights are
the seeded synthetic state dict (nuhtc_amd.weights.seeded_state_dict) that tests regenerate on
any box, inputs are synthetic nuclei tiles (nuhtc_amd.synth).  The float NCHW network input is
produced by oracle.model.preprocess because cv2 (the reference's resize) is absent here: the
reference boundary for these goldens is `model.simple_test(img, img_metas, rescale=True)`.

Large tensors are stored as a fixed-stride subsample plus float64 sum / abs-sum.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nuhtc_amd import synth, weights  # noqa: E402
from oracle import model as O  # noqa: E402
from oracle.ref_harness import mmcv_stub  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')
CFG = '/root/reference/configs/nuhtc/htc_lite_swin_pytorch_fpn_PanNuke_seasaw_CAS.py'
MAXN = 8192
# synthetic weights give near-uniform class logits; this fixed offset on every stage's fc_cls.bias puts two
# classes around the 0.35 score threshold and objectness high, so NMS/labels/masks all get exercised
CLS_TARGET = [1.2, 1.2, 0.0, -0.5, -0.3, 3.0, 0.0]


def calibrate(seed, tiles, channel_mode):
    """Pick the two synthetic-weight offsets from one oracle pass: per-stage fc_cls bias offset that moves the
    mean ensemble logits to CLS_TARGET, and a semantic-logit bias that leaves ~30 % foreground."""
    sd = weights.seeded_state_dict(seed)
    _, it = O.Oracle(sd)(tiles, channel_mode, keep=True)
    cm = (sum(it['stage_cls']) / 3).mean(0).numpy()
    cls_add = (np.array(CLS_TARGET, np.float32) - cm).astype(np.float32)
    sem_bias = float(sd['roi_head.semantic_head.conv_logits.bias'][0]) - float(np.quantile(it['sem_pred'].numpy(), 0.7))
    return cls_add, np.float32(sem_bias)


def five_class_head(seed, tiles, channel_mode, sem_bias, target_dets=60):
    """Head weights for the five-class case.  The seeded N(0, sigma) classifier rows are nearly parallel to nothing in
    particular, so class logits hardly vary from RoI to RoI and at most two classes ever pass the 0.35 threshold.  Here the
    five class rows of every stage's NormedLinear are set to principal directions of that stage's (normalised) fc features
    over the RoIs of these tiles -- stage 1/2 directions matched (with sign) to the stage-0 ones by correlation, so the three
    stages vote alike -- the class biases centre each class logit, and one objectness offset is bisected (on the oracle's
    detection post-processing) to ~target_dets detections per tile.  Returns {name: tensor} overrides, stored in the fixture."""
    import torch.nn.functional as F
    sd = weights.seeded_state_dict(seed)
    sd['roi_head.semantic_head.conv_logits.bias'] = torch.tensor([float(sem_bias)])
    _, it = O.Oracle(sd)(tiles, channel_mode, keep=True)
    feats = []
    for k in range(3):
        p = f'roi_head.bbox_head.{k}.'
        f = O.bbox_feats(it['x'], it['sem_feat'], it['stage_rois'][k]).flatten(1)
        h = F.relu(F.linear(f, sd[p + 'shared_fcs.0.weight'], sd[p + 'shared_fcs.0.bias']))
        h = F.relu(F.linear(h, sd[p + 'shared_fcs.1.weight'], sd[p + 'shared_fcs.1.bias']))
        feats.append(h)

    def pcs(h, n):
        xh = h / (h.norm(dim=1, keepdim=True) + 1e-6)
        _, _, vt = torch.linalg.svd(xh - xh.mean(0, keepdim=True), full_matrices=False)
        return xh, vt[:n]
    xh0, v0 = pcs(feats[0], 5)
    s0 = (xh0 @ v0.T).numpy()
    over = {}
    rows = [v0]
    for k in (1, 2):
        # stage-k directions: ridge regression of the stage-0 class scores on the stage-k features (the best linear
        # agreement between the stages' votes)
        xh, _ = pcs(feats[k], 1)
        xc = (xh - xh.mean(0, keepdim=True)).double()
        gram = xc.T @ xc
        lam = 1e-3 * float(torch.trace(gram)) / gram.shape[0]
        u = torch.linalg.solve(gram + lam * torch.eye(gram.shape[0], dtype=torch.float64), xc.T @ torch.from_numpy(s0).double())
        u = (u / u.norm(dim=0, keepdim=True)).T.float()
        for c in range(5):
            print('stage', k, 'class', c, 'corr with stage 0', round(float(np.corrcoef(s0[:, c], (xh @ u[c]).numpy())[0, 1]), 3))
        rows.append(u)
    # equal spread for the five classes (else the first principal direction wins almost every RoI): mix each row with the
    # mean feature direction, which lowers the spread of cos(x, row) without changing which RoIs rank high
    for k in range(3):
        xh = feats[k] / (feats[k].norm(dim=1, keepdim=True) + 1e-6)
        m = xh.mean(0)
        m = m / m.norm()

        def spread(row, a):
            w = a * m + row
            return float((xh @ (w / w.norm())).std())
        target = min(spread(r, 0.0) for r in rows[k])
        mixed = []
        for r in rows[k]:
            lo, hi = 0.0, 50.0
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                if spread(r, mid) > target:
                    lo = mid
                else:
                    hi = mid
            mixed.append(0.5 * (lo + hi) * m + r)
        rows[k] = torch.stack(mixed)
    for k in range(3):
        p = f'roi_head.bbox_head.{k}.'
        w = sd[p + 'fc_cls.weight'].clone()
        w[:5] = rows[k]
        b = sd[p + 'fc_cls.bias'].clone()
        sd[p + 'fc_cls.weight'] = w
        cls, _ = O.bbox_head(sd, k, O.bbox_feats(it['x'], it['sem_feat'], it['stage_rois'][k]))
        b[:5] = b[:5] - cls[:, :5].mean(0)
        over[p + 'fc_cls.weight'] = w
        over[p + 'fc_cls.bias'] = b
        sd[p + 'fc_cls.bias'] = b
    stage = [O.bbox_head(sd, k, O.bbox_feats(it['x'], it['sem_feat'], it['stage_rois'][k])) for k in range(3)]
    cls_mean = sum(c for c, _ in stage) / 3.0
    rois, reg = it['stage_rois'][2], stage[2][1]
    img_hw = tuple(O.preprocess(tiles[:1], channel_mode).shape[-2:])
    nb = len(tiles)

    def per_class(db, off):
        c = cls_mean.clone()
        c[:, :5] += db
        c[:, 5] += off
        lab = torch.cat([O.detect_post(rois[rois[:, 0] == i, 1:], c[rois[:, 0] == i], reg[rois[:, 0] == i], img_hw, 2.0)[1]
                         for i in range(nb)])
        return np.bincount(lab.numpy(), minlength=5).astype(np.float64)

    def bisect(db):
        lo, hi = -12.0, 12.0
        for _ in range(20):
            mid = 0.5 * (lo + hi)
            if per_class(db, mid).sum() > target_dets * nb:
                hi = mid
            else:
                lo = mid
        return 0.5 * (lo + hi)
    # balance the classes: one objectness offset sets the total (~target_dets per tile), per-class bias shifts even out
    # the detections per class; alternate a few times
    db = torch.zeros(5)
    for it_ in range(40):
        off = bisect(db)
        n = per_class(db, off)
        print('balance', it_, 'offset', round(off, 3), 'dets per class', n)
        db = db + torch.from_numpy((0.3 * 0.93 ** it_) * np.clip(np.log((n.mean() + 1) / (n + 1)), -1, 1)).float()
    db = torch.from_numpy(np.round(db.numpy(), 3))
    off = np.float32(round(bisect(db), 3))
    for k in range(3):
        p = f'roi_head.bbox_head.{k}.fc_cls.bias'
        over[p] = over[p].clone()
        over[p][:5] += db
        over[p][5] += float(off)
    count = lambda o: per_class(db, o).sum() / nb
    print('five-class head: objectness offset', off, 'dets/tile', count(float(off)))
    return over


def sub(t):
    """subsample + checksums of a tensor (shared with tests/golden_util.py)."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    flat = a.reshape(-1).astype(np.float32)
    stride = max(1, -(-flat.size // MAXN))
    return dict(shape=np.array(a.shape), sample=flat[::stride].copy(), stride=np.array(stride),
                sum=np.array(flat.astype(np.float64).sum()), asum=np.array(np.abs(flat.astype(np.float64)).sum()))


def put(store, name, t):
    for k, v in sub(t).items():
        store[f'{name}.{k}'] = v


def run_case(name, tile_size, n_tiles, seed, channel_mode, five=False):
    model, cfg = mmcv_stub.build_reference_detector(CFG)
    if isinstance(tile_size, tuple):      # (h, w) not multiples of 16: cut from the next larger square tile; Pad(size_divisor=32) acts
        th, tw = tile_size
        tiles = np.ascontiguousarray(synth.nuclei_tiles(n_tiles, -(-max(th, tw) // 32) * 32, start=100 * seed)[:, :th, :tw])
    else:
        th = tw = tile_size
        tiles = synth.nuclei_tiles(n_tiles, tile_size, start=100 * seed)
    CLS_BIAS_ADD, sem_bias = calibrate(seed, tiles, channel_mode)
    sd = weights.seeded_state_dict(seed)
    sd['roi_head.semantic_head.conv_logits.bias'] = torch.tensor([float(sem_bias)])
    overrides = {}
    if five:
        CLS_BIAS_ADD = np.zeros(7, np.float32)
        overrides = five_class_head(seed, tiles, channel_mode, sem_bias)
        sd.update(overrides)
    for k in range(3):
        sd[f'roi_head.bbox_head.{k}.fc_cls.bias'] = sd[f'roi_head.bbox_head.{k}.fc_cls.bias'] + torch.from_numpy(CLS_BIAS_ADD)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(('relative_position_index', 'cum_samples')) or k == 'roi_head.kernel' for k in missing), missing
    # reference schema == our schema (names and shapes), pinned here
    ref_sd = model.state_dict()
    for k, v in sd.items():
        assert tuple(ref_sd[k].shape) == tuple(v.shape), k

    img = O.preprocess(tiles, channel_mode)
    Hn, Wn = img.shape[-2:]
    # Resize(scale_factor=2, keep_ratio=True) -> mmcv.imrescale: new size int(size * 2 + 0.5), scale_factor recomputed = 2 exactly;
    # img_shape is the resized image, pad_shape the tensor after Pad(size_divisor=32) (transforms.py:207-236,570-)
    metas = [dict(img_shape=(2 * th, 2 * tw, 3), ori_shape=(th, tw, 3), pad_shape=(Hn, Wn, 3),
                  scale_factor=np.array([2, 2, 2, 2], np.float32), flip=False, flip_direction=None)
             for _ in range(n_tiles)]
    cap = {}

    def hook(key):
        def f(mod, inp, out):
            cap.setdefault(key, []).append((inp, out))
        return f
    model.backbone.register_forward_hook(hook('backbone'))
    model.neck.register_forward_hook(hook('neck'))
    model.roi_head.semantic_head.register_forward_hook(hook('sem'))
    for k in range(3):
        model.roi_head.bbox_head[k].register_forward_hook(hook(f'bbox{k}'))
    model.roi_head.mask_head[0].register_forward_hook(hook('mask'))
    for s in range(4):
        for b, blk in enumerate(model.backbone.stages[s].blocks):
            blk.register_forward_hook(hook(f's{s}b{b}'))
    model.backbone.patch_embed.register_forward_hook(hook('embed'))

    orig_rpn = model.rpn_head.simple_test_rpn

    def rpn_wrap(x, m):
        cap['rpn_convs'] = model.rpn_head(x)
        out = orig_rpn(x, m)
        cap['rpn_props'] = [o.clone() for o in out]
        return out
    model.rpn_head.simple_test_rpn = rpn_wrap
    orig_ws = model.roi_head._watershed_proposal

    def ws_wrap(*a, **k):
        pl, wl = orig_ws(*a, **k)
        cap['ws'] = [w.clone() for w in wl]
        return pl, wl
    model.roi_head._watershed_proposal = ws_wrap

    with torch.no_grad():
        results = model.simple_test(img, metas, rescale=True)

    g = dict(tiles=tiles, seed=np.array(seed), channel_mode=np.array(channel_mode), sem_bias=np.array(sem_bias, np.float32),
             cls_bias_add=np.array(CLS_BIAS_ADD, np.float32))
    for k, v in overrides.items():
        g['override.' + k] = v.numpy()
    put(g, 'embed', cap['embed'][0][1][0])
    for s in range(4):
        for b in range(O.DEPTHS[s]):
            put(g, f's{s}b{b}', cap[f's{s}b{b}'][0][1])
    for i, t in enumerate(cap['backbone'][0][1]):
        put(g, f'c{i}', t)
    for i, t in enumerate(cap['neck'][0][1]):
        put(g, f'x{i}', t)
    for i in range(4):
        put(g, f'rpn_cls{i}', cap['rpn_convs'][0][i])
        put(g, f'rpn_reg{i}', cap['rpn_convs'][1][i])
    put(g, 'sem_pred', cap['sem'][0][1][0])
    put(g, 'sem_feat', cap['sem'][0][1][1])
    for i in range(n_tiles):
        g[f'rpn_props{i}'] = cap['rpn_props'][i].numpy()
        g[f'ws{i}'] = cap['ws'][i].numpy()
    for k in range(3):
        inp, out = cap[f'bbox{k}'][0]
        put(g, f'bbox_feats{k}', inp[0])
        g[f'cls{k}'] = out[0].numpy()
        g[f'reg{k}'] = out[1].numpy()
    if 'mask' in cap:
        inp, out = cap['mask'][0]
        put(g, 'mask_feats', inp[0])
        put(g, 'mask_logits', out[0])
    for i, (br, sr) in enumerate(results):
        g[f'det{i}'] = np.concatenate(br, 0).astype(np.float32)
        g[f'lab{i}'] = np.concatenate([np.full(len(b), c, np.int32) for c, b in enumerate(br)])
        ms = [m for cl in sr for m in cl]
        g[f'masks{i}'] = np.packbits(np.stack(ms).astype(np.uint8), axis=-1) if ms else np.zeros((0, th, -(-tw // 8)), np.uint8)
        print(name, 'tile', i, 'rpn', len(cap['rpn_props'][i]), 'ws', len(cap['ws'][i]), 'dets', len(g[f'det{i}']),
              'classes', np.bincount(g[f'lab{i}'], minlength=5))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **g)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


CASES = [
    # name, tile_size, n_tiles, seed, channel_mode
    ('small_b2', 64, 2, 0, 0),
    ('small_wsi_b3', 96, 3, 1, 1),
    ('full_b1', 256, 1, 2, 0),
    # all five classes live, ~60 detections per 256x256 tile (a realistic PanNuke load), WSI channel mode, batch of 2
    ('five_b2', 256, 2, 3, 1, True),
    # an image whose resized size is not a multiple of 32: 72 x 90 -> img_shape 144 x 180 -> pad_shape 160 x 192 (Pad acts;
    # the reference clips boxes to img_shape, squeezes the padded semantic map into img_shape for the component proposals and
    # pastes masks into ori_shape)
    ('pad_b2', (72, 90), 2, 4, 0),
]

if __name__ == '__main__':
    only = sys.argv[1:]
    for c in CASES:
        if not only or c[0] in only:
            run_case(*c)
