"""GPU parity of the whole tile-inference path (proposals, cascade, detections, masks, per-tile mask-NMS) against
the oracle and the committed reference goldens, through the C ABI."""
import numpy as np
import pytest
import torch

import golden_util as G
import parity_util as P

pytestmark = pytest.mark.gpu


def _engine(g, **kw):
    from nuhtc_amd.engine import Engine
    sd = G.seeded_sd(g)
    tiles = g['tiles']
    return Engine(sd, device=0, max_batch=kw.pop('max_batch', len(tiles)), tile=tiles.shape[1:3], **kw), sd


def test_roi_align_op(hip_device):
    from oracle import ops
    g = G.load('small_b2')
    eng, _ = _engine(g)
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((2, 64, 24, 20)).astype(np.float32)
    xy = rng.uniform(-12, 90, (200, 2)); wh = rng.uniform(0.0, 70, (200, 2))
    rois = np.concatenate([rng.integers(0, 2, (200, 1)), xy, xy + wh], 1).astype(np.float32)
    rois[0, 1:] = [5, 5, 5, 5]          # zero-size roi
    rois[1, 1:] = [-50, -50, -40, -40]  # fully outside
    f_nhwc = torch.from_numpy(feat).permute(0, 2, 3, 1).contiguous().cuda()
    for P, sr, sc in [(7, 2, 0.25), (14, 0, 0.25), (7, 2, 0.125), (14, 0, 0.125)]:
        ref = ops.roi_align(feat, rois, P, sc, sr)                       # (R,C,P,P)
        out = eng.op_roi_align(f_nhwc, torch.from_numpy(rois).cuda(), P, sc, sr).cpu().numpy()   # (R,P,P,C)
        err = np.abs(out.transpose(0, 3, 1, 2) - ref).max()
        print('roi_align', P, sr, sc, 'max err', err)
        assert err <= 1e-5


def test_nms_op(hip_device):
    from oracle import ops
    g = G.load('small_b2')
    eng, _ = _engine(g)
    rng = np.random.default_rng(1)
    for n in (1, 63, 64, 65, 700, 5000, 12000):
        xy = rng.uniform(0, 500, (n, 2)); wh = rng.uniform(5, 80, (n, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        scores = np.round(rng.uniform(0, 1, n), 3).astype(np.float32)   # plenty of exact ties
        for thr in (0.3, 0.7):
            ref = ops.nms(boxes, scores, thr)
            got = eng.op_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), thr).cpu().numpy()
            assert len(ref) == len(got) and (ref == got).all(), (n, thr, len(ref), len(got))


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize('case', ['small_b2', 'small_wsi_b3', 'full_b1', 'five_b2'])
def test_full_path_vs_oracle_and_golden(hip_device, case):
    """Stage-by-stage parity with the oracle fed the engine's own inputs of that stage (so rounding differences do
    not compound through the thresholded / greedy steps), then end-to-end agreement with oracle and reference golden."""
    from oracle import model as O
    g = G.load(case)
    eng, sd = _engine(g)
    eng.enable_token_dump()
    tiles = g['tiles']
    B = len(tiles)
    mode = int(g['channel_mode'])
    eng.infer_async(eng.to_device(tiles), mode)
    got = eng.results(B)
    nc = eng.cfg.num_classes
    Hn, Wn = 2 * tiles.shape[1], 2 * tiles.shape[2]
    msgs = []

    def check(name, err, tol):
        print(f'  {name}: {err:.3e} (tol {tol:.1e})')
        if not err <= tol:
            msgs.append(f'{name}: {err} > {tol}')

    x = [_nchw(eng.buffer(f'x{i}')[:B]) for i in range(4)]
    sem_feat = _nchw(eng.buffer('sem_feat')[:B])
    sem_pred = eng.buffer('sem_pred')[:B].cpu()[:, None]
    # --- a10-a12: RPN proposals from the engine's own RPN maps
    rp = [_nchw(eng.buffer(f'rpn{i}')[:B]) for i in range(4)]
    rpn_ref = O.rpn_proposals([r[:, 0:3] for r in rp], [r[:, 3:15] for r in rp], (Hn, Wn))
    rpn_counts = eng.buffer('rpn_counts')[:B].cpu().numpy()
    rpn = eng.buffer('rpn_props')[:B].cpu().numpy()
    for i in range(B):
        a, b = rpn_ref[i].numpy(), rpn[i, :rpn_counts[i]]
        print(case, 'tile', i, 'rpn proposals oracle/hip', len(a), len(b))
        if len(a) != len(b):
            msgs.append(f'tile {i}: rpn count {len(b)} != {len(a)}')
        else:
            # same multiset of proposals; the order may swap where two scores agree to ~1 ulp (expf vs torch sigmoid)
            d = np.abs(a[:, None, :] - b[None, :, :]).max(-1)
            displaced = int((np.abs(a - b).max(1) > 1e-3).sum())
            lone_a, lone_b = np.nonzero(d.min(1) > 1e-3)[0], np.nonzero(d.min(0) > 1e-3)[0]
            print(f'  tile {i} rpn proposals: {len(lone_a)} oracle rows / {len(lone_b)} hip rows without a partner, {displaced} rows at a different rank')
            for rows, other, lone, tag in ((a, b, lone_a, 'oracle'), (b, a, lone_b, 'hip')):
                for k in lone:
                    why = P.explain_proposal(rows[k], other)
                    print(f'    {tag}-only proposal {rows[k].tolist()}: {why}')
                    if why is None:
                        msgs.append(f'tile {i}: {tag}-only rpn proposal {rows[k].tolist()} is not explained by a threshold')
    # --- a14: connected-component proposals from the engine's own semantic logits: exact
    cc_ref = O.cc_proposals(sem_pred, (Hn, Wn))
    cc_counts = eng.buffer('cc_counts')[:B].cpu().numpy()
    cc = eng.buffer('cc_props')[:B].cpu().numpy()
    for i in range(B):
        a, b = cc_ref[i].numpy()[:, :4], cc[i, :cc_counts[i]]
        print(case, 'tile', i, 'cc proposals oracle/hip', len(a), len(b))
        if a.shape != b.shape or (a != b).any():
            msgs.append(f'tile {i}: cc proposals differ: {a.shape} vs {b.shape}')
    # --- a15: roi list = cat(cc, rpn) per image
    R = int(eng.buffer('roi_total').item())
    roi_cnt = eng.buffer('roi_counts')[:B].cpu().numpy()
    assert R == int((rpn_counts + cc_counts).sum()) and (roi_cnt == rpn_counts + cc_counts).all()
    # --- a16-a19: cascade stages, each fed the engine's rois of that stage
    stage_rois = [eng.buffer(f'rois_stage{k}')[:R].cpu() for k in range(3)]
    cls_hip = [eng.buffer(f'cls{k}')[:R, :nc + 2].cpu() for k in range(3)]
    reg_hip = [eng.buffer(f'reg{k}')[:R].cpu() for k in range(3)]
    with torch.no_grad():
        for k in range(3):
            feats = O.bbox_feats(x, sem_feat, stage_rois[k])
            if k == 2:
                f_hip = eng.buffer('bbox_feats')[:R].cpu().reshape(R, 7, 7, 64).permute(0, 3, 1, 2)
                check('stage 2 roi features', float((f_hip - feats).abs().max()), 2e-4)
            cls, reg = O.bbox_head(sd, k, feats)
            check(f'stage {k} cls', float((cls - cls_hip[k]).abs().max()), 1e-3)
            check(f'stage {k} reg', float((reg - reg_hip[k]).abs().max()), 1e-3)
            if k < 2:
                ref_next = O.delta2bbox(stage_rois[k][:, 1:], reg_hip[k], O.STAGE_STDS[k], (Hn, Wn))
                check(f'stage {k} refined rois', float((ref_next - stage_rois[k + 1][:, 1:]).abs().max()), 1e-3)
        # --- a20-a22: ensemble + Seesaw + multiclass NMS from the engine's logits / deltas
        cls_mean = (cls_hip[0] + cls_hip[1] + cls_hip[2]) / 3.0
        off = 0
        hip_boxes = eng.boxes[:B].cpu().numpy()
        hip_labels = eng.labels[:B].cpu().numpy()
        hip_counts = eng.counts[:B].cpu().numpy()
        for i in range(B):
            sl = slice(off, off + int(roi_cnt[i])); off += int(roi_cnt[i])
            d, l = O.detect_post(stage_rois[2][sl, 1:], cls_mean[sl], reg_hip[2][sl], (Hn, Wn), 2.0)
            n = int(hip_counts[i])
            print(case, 'tile', i, 'detections oracle/hip', len(d), n)
            if len(d) != n:
                msgs.append(f'tile {i}: det count {n} != {len(d)}')
                continue
            if n:
                check(f'tile {i} det rows (NMS order)', float(np.abs(d.numpy() - hip_boxes[i, :n]).max()), 1e-3)
                if (l.numpy() != hip_labels[i, :n]).any():
                    msgs.append(f'tile {i}: labels differ')
        # --- a23-a26: mask branch from the engine's detections
        D = int(eng.buffer('det_total').item())
        assert D == int(hip_counts.sum())
        mrois = eng.buffer('mask_rois')[:D].cpu()
        if D:
            mf = O.roi_extract(x, mrois, 14, 0) + O.semantic_roi(sem_feat, mrois)
            mf_hip = eng.buffer('mask_feats')[:D].cpu().reshape(D, 14, 14, 64).permute(0, 3, 1, 2)
            check('mask roi features', float((mf - mf_hip).abs().max()), 2e-4)
            prob = O.mask_head(sd, mf)
            prob_hip = eng.buffer('mask_prob')[:D].cpu()[:, None]
            check('mask probabilities', float((prob - prob_hip).abs().max()), 1e-4)
            off = 0
            for i in range(B):
                n = int(hip_counts[i])
                pm = O.paste_masks(prob_hip[off:off + n], (mrois[off:off + n, 1:] / 2.0), tiles.shape[1], tiles.shape[2])
                off += n
                hm = np.stack([m for cl in got[i][1] for m in cl]) if n else np.zeros((0,) + tiles.shape[1:3], bool)
                # got[i] is class-grouped; regroup the oracle paste the same way
                lab = hip_labels[i, :n]
                order = np.concatenate([np.nonzero(lab == c)[0] for c in range(nc)]) if n else np.zeros(0, int)
                diff = pm[order] != hm
                nd = int(diff.sum())
                print(case, 'tile', i, 'pasted mask pixel mismatches', nd, 'of', int(hm.sum()), 'set pixels')
                if nd:
                    # a pixel may differ only where the pasted probability sits on the 0.5 threshold
                    _, vals = O.paste_masks(prob_hip[off - n:off], (mrois[off - n:off, 1:] / 2.0), tiles.shape[1], tiles.shape[2], return_values=True)
                    dist = float(np.abs(vals[order][diff] - 0.5).max())
                    print(f'    farthest differing pixel: pasted probability {dist:.2e} from 0.5')
                    if dist > 1e-5:
                        msgs.append(f'tile {i}: {nd} pasted mask pixels differ, up to {dist:.2e} from the threshold')
                areas = eng.areas[i, :n].cpu().numpy()
                if n and (areas[order] != hm.reshape(n, -1).sum(1)).any():
                    msgs.append(f'tile {i}: mask areas differ from popcounts')
    # --- a27-a28: per-tile filter + mask-NMS (tools/infer_wsi.py:510-531) on the engine's results
    keep = eng.keep[:B].cpu().numpy()
    for i in range(B):
        kb, kl, km = O.tile_filter_and_mask_nms(got[i][0], got[i][1], size=tiles.shape[1], margin=2, min_area=10, thr=0.05)
        n = int(hip_counts[i])
        hip_kept = hip_boxes[i, :n][keep[i, :n] == 1]
        a = G.canon_rows(kb) if len(kb) else kb
        b = G.canon_rows(hip_kept) if len(hip_kept) else hip_kept
        print(case, 'tile', i, 'mask-nms kept oracle/hip', len(kb), len(hip_kept))
        if a.shape != b.shape or (len(a) and np.abs(a - b).max() > 0):
            msgs.append(f'tile {i}: mask-NMS keep set differs ({len(kb)} vs {len(hip_kept)})')
    # --- end to end: free-running oracle and the reference golden (rounding differences may flip near-threshold
    # decisions; report the agreement, require it to be near-total)
    ref, it = O.Oracle(sd)(tiles, mode, keep=True)
    vals = P.oracle_paste_values(O, it, tiles.shape[1:3])
    for i in range(B):
        gd, gl = g[f'det{i}'], g[f'lab{i}']
        gm = np.unpackbits(g[f'masks{i}'], axis=-1).astype(bool)
        gold = ([gd[gl == c] for c in range(5)], [[gm[j] for j in range(len(gd)) if gl[j] == c] for c in range(5)])
        # the golden was produced by the reference's own code on the build container's CPU, the oracle runs on this box's CPU:
        # they list the same instances in the same order unless a decision flipped between the two CPUs
        ob = P.flatten(ref[i])[0]
        same_order = len(gd) == len(ob) and (len(gd) == 0 or np.abs(gd - ob).max() < 1e-3)
        for tag, r, v in (('oracle', ref[i], vals[i]), ('golden', gold, vals[i] if same_order else None)):
            rep, fails = P.compare_strict(r, got[i], values=v, values_side='ref')
            print(f'{case} tile {i} end-to-end vs {tag}: {P.fmt(rep)}')
            for e in rep['explained']:
                print('    tolerated:', e)
            msgs += [f'tile {i} vs {tag}: {f}' for f in fails]
    assert not msgs, '\n'.join(msgs)


def test_fp32_mfma_pipe_end_to_end(hip_device):
    """matrix_pipe = NUHTC_PIPE_FP32 (every matrix product on v_mfma_f32_32x32x2_f32, round 1's kernels) against the oracle with the
    same strict gates, and against the default pipe (exact bf16 split): same instances, same classes, same masks -- both pipes are
    fp32 arithmetic, so they may only differ where the oracle comparison tolerates it.  An unknown pipe value is refused."""
    from nuhtc_amd import hip
    from nuhtc_amd.engine import HipError
    from oracle import model as O
    g = G.load('five_b2')
    tiles = g['tiles']
    B, mode = len(tiles), int(g['channel_mode'])
    e32, sd = _engine(g, matrix_pipe=hip.PIPE_FP32)
    esp, _ = _engine(g)
    assert esp.cfg.matrix_pipe == hip.PIPE_BF16_SPLIT
    e32.infer_async(e32.to_device(tiles), mode)
    esp.infer_async(esp.to_device(tiles), mode)
    got32, gotsp = e32.results(B), esp.results(B)
    ref, it = O.Oracle(sd)(tiles, mode, keep=True)
    vals = P.oracle_paste_values(O, it, tiles.shape[1:3])
    msgs = []
    for i in range(B):
        rep, fails = P.compare_strict(ref[i], got32[i], values=vals[i], values_side='ref')
        print(f'tile {i} fp32 pipe vs oracle: {P.fmt(rep)}')
        msgs += [f'tile {i} fp32 pipe vs oracle: {f}' for f in fails]
        rep, fails = P.compare_strict(gotsp[i], got32[i], values=None)
        print(f'tile {i} fp32 pipe vs split pipe: {P.fmt(rep)}')
        if rep['n_ref'] != rep['n_got'] or rep['matched'] != rep['n_ref']:
            msgs.append(f'tile {i}: pipes disagree on the instances: {P.fmt(rep)}')
    assert not msgs, '\n'.join(msgs)
    with pytest.raises(HipError):
        _engine(g, matrix_pipe=7)


def test_capacity_overflow_is_reported(hip_device):
    g = G.load('full_b1')
    eng, _ = _engine(g, max_cc_proposals=2)
    eng.infer_async(eng.to_device(g['tiles']), 0)
    from nuhtc_amd.engine import HipError
    with pytest.raises(HipError):
        eng.check()


def test_consep_classes_and_batch_independence(hip_device):
    """CoNSeP-style head (num_classes=4, max_per_img=300) at full tile size, B=6 in a max_batch=16 engine: every tile
    must equal the oracle, and a tile's result must not depend on its position in the batch or on its neighbours."""
    from nuhtc_amd import synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(5, num_classes=4, obj_bias=0.5)
    eng = Engine(sd, device=0, max_batch=16, tile=(256, 256), num_classes=4, max_per_img=300)
    base = synth.nuclei_tiles(3, 256, start=40)
    tiles = np.stack([base[0], base[1], base[2], base[0], base[2], base[1]])
    got = eng(tiles, 1)
    ref, it = O.Oracle(sd, num_classes=4, max_per_img=300)(base, 1, keep=True)
    vals = P.oracle_paste_values(O, it, (256, 256))
    msgs = []
    for i, r in zip((0, 1, 2), ref):
        rep, fails = P.compare_strict(r, got[i], max_per_img=300, values=vals[i])
        print(f'consep tile {i}: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        assert rep['n_got'] <= 300
        msgs += [f'tile {i}: {f}' for f in fails]
    for a, b in ((0, 3), (2, 4), (1, 5)):   # same tile at different batch positions -> bit-identical outputs
        for c in range(4):
            if not np.array_equal(got[a][0][c], got[b][0][c]) or any(not np.array_equal(x, y) for x, y in zip(got[a][1][c], got[b][1][c])):
                msgs.append(f'tiles {a} and {b} differ in class {c}')
    assert not msgs, '\n'.join(msgs)


def test_roi_features_all_size_classes(hip_device):
    """7x7 RoI features across the three code paths (square LDS tiles, row bands, per-bin gathers) and the 14x14 mask features:
    RoIs from 8 to 300 px with independent sides (so the semantic sample grid differs per axis), some hanging over the image
    border, given to the engine through the fixed-load entry point; the oracle is fed the engine's own maps and stage RoIs.
    Tolerance: fp32 sums in a different order (x0 + sem interpolated together, packed accumulation): 2e-4 absolute."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(3)
    B, n = 2, 480
    eng = Engine(sd, device=0, max_batch=B, tile=(256, 256))
    eng.enable_token_dump()            # (also keeps the per-stage RoI lists)
    rng = np.random.default_rng(17)
    wh = np.concatenate([rng.uniform(8, 44, (n // 3, 2)), rng.uniform(40, 115, (n // 3, 2)), rng.uniform(100, 300, (n // 3, 2))])
    wh = np.stack([rng.permutation(wh) for _ in range(B)]).astype(np.float32)
    ctr = rng.uniform(-10, 522, (B, n, 2)).astype(np.float32)
    rois = np.clip(np.concatenate([ctr - wh / 2, ctr + wh / 2], -1), 0, 512).astype(np.float32)
    rois[:, :, 2:] = np.maximum(rois[:, :, 2:], rois[:, :, :2] + 4)           # keep them non-degenerate
    tiles = eng.to_device(synth.nuclei_tiles(B, 256, start=3))
    eng.infer_fixed_load_async(tiles, torch.from_numpy(rois).to(tiles.device), 40, hip.CH_SWAP)
    eng.check()
    R = B * n
    counts = eng.buffer('roi_fallback_count').cpu().numpy()
    assert counts[0] > 20 and counts[1] > 20, counts        # big and mid-size classes are both populated
    x = [_nchw(eng.buffer(f'x{i}')[:B]) for i in range(4)]
    sem_feat = _nchw(eng.buffer('sem_feat')[:B])
    r2 = eng.buffer('rois_stage2')[:R].cpu()
    with torch.no_grad():
        ref = O.bbox_feats(x, sem_feat, r2)
    got = eng.buffer('bbox_feats')[:R].cpu().reshape(R, 7, 7, 64).permute(0, 3, 1, 2)
    err = (got - ref).abs().reshape(R, -1).max(1).values
    w = (r2[:, 3] - r2[:, 1]).numpy(); h = (r2[:, 4] - r2[:, 2]).numpy()
    for lo, hi in ((0, 44), (44, 112), (112, 1000)):
        sel = (np.maximum(w, h) >= lo) & (np.maximum(w, h) < hi)
        print(f'RoIs with max side in [{lo},{hi}): {int(sel.sum())}, max |err| {float(err[torch.from_numpy(sel)].max()) if sel.any() else 0:.2e}')
    assert float(err.max()) <= 2e-4, float(err.max())
    # 14x14 mask features of the 40 detections per tile (boxes of all sizes as well)
    D = int(eng.buffer('det_total').item())
    assert D == B * 40
    mrois = eng.buffer('mask_rois')[:D].cpu()
    with torch.no_grad():
        mref = O.roi_extract(x, mrois, 14, 0) + O.semantic_roi(sem_feat, mrois)
    mgot = eng.buffer('mask_feats')[:D].cpu().reshape(D, 14, 14, 64).permute(0, 3, 1, 2)
    ms = (mrois[:, 3:] - mrois[:, 1:3]).max(1).values
    print(f'mask RoIs: {D}, max side {float(ms.min()):.0f}..{float(ms.max()):.0f} px, max |err| {float((mgot - mref).abs().max()):.2e}')
    assert float((mgot - mref).abs().max()) <= 2e-4


def test_roi_features_giant_boxes_at_1024_px(hip_device):
    """Boxes beyond the big-box kernel's tables (more than 24 samples per bin and axis: sides over ~670 px, which only a network input
    above 512 px can hold -- 20x slides, scale_factor 4) take mmcv's own sample loop (roi_feat7_giant_kernel); the big-box kernel right
    below that limit, with its widest bins.  Engine maps and stage RoIs against the oracle, 2e-4 absolute."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(3)
    B, n = 1, 96
    eng = Engine(sd, device=0, max_batch=B, tile=(256, 256), scale_factor=4.0)
    eng.enable_token_dump()
    rng = np.random.default_rng(23)
    wh = np.concatenate([rng.uniform(120, 660, (n // 2, 2)), rng.uniform(680, 1020, (n // 2, 2))]).astype(np.float32)[None]
    ctr = rng.uniform(100, 924, (B, n, 2)).astype(np.float32)
    rois = np.clip(np.concatenate([ctr - wh / 2, ctr + wh / 2], -1), 0, 1024).astype(np.float32)
    tiles = eng.to_device(synth.nuclei_tiles(B, 256, start=9))
    eng.infer_fixed_load_async(tiles, torch.from_numpy(rois).to(tiles.device), 8, hip.CH_SWAP)
    eng.check()
    counts = eng.buffer('roi_fallback_count').cpu().numpy()
    assert counts[0] > 10 and counts[2] > 10, counts        # big-box and giant classes are both populated
    x = [_nchw(eng.buffer(f'x{i}')[:B]) for i in range(4)]
    sem_feat = _nchw(eng.buffer('sem_feat')[:B])
    r2 = eng.buffer('rois_stage2')[:n].cpu()
    with torch.no_grad():
        ref = O.bbox_feats(x, sem_feat, r2)
    got = eng.buffer('bbox_feats')[:n].cpu().reshape(n, 7, 7, 64).permute(0, 3, 1, 2)
    err = (got - ref).abs().reshape(n, -1).max(1).values
    side = (r2[:, 3:] - r2[:, 1:3]).max(1).values
    print(f'boxes {float(side.min()):.0f}..{float(side.max()):.0f} px: big {int(counts[0])}, giant {int(counts[2])}, max |err| {float(err.max()):.2e}')
    assert float(err.max()) <= 2e-4, float(err.max())


@pytest.mark.gpu
def test_roi_features_long_lists_and_batch_independence(hip_device):
    """The mid-size and big-box RoI kernels have a short-list and a long-list form (roi.hip: roi_feat7_stream_few_kernel / one
    workgroup per map below 1024 / 512 boxes per batch, roi_feat7_stream_kernel / one workgroup per box above).  Here the lists are
    long -- 1400 RoIs of 44-300 px per tile -- and checked against the oracle on a sample; then the first 150 RoIs of every tile are
    run alone (short lists): their features must be the same BIT FOR BIT, a box's result may not depend on what else the batch holds."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(3)
    B, n, m = 2, 1400, 150
    eng = Engine(sd, device=0, max_batch=B, tile=(256, 256))
    eng.enable_token_dump()
    rng = np.random.default_rng(23)
    wh = np.concatenate([rng.uniform(44, 112, (n // 2, 2)), rng.uniform(112, 300, (n // 2, 2))])
    wh = np.stack([rng.permutation(wh) for _ in range(B)]).astype(np.float32)
    ctr = rng.uniform(0, 512, (B, n, 2)).astype(np.float32)
    rois = np.clip(np.concatenate([ctr - wh / 2, ctr + wh / 2], -1), 0, 512).astype(np.float32)
    rois[:, :, 2:] = np.maximum(rois[:, :, 2:], rois[:, :, :2] + 4)
    tiles = eng.to_device(synth.nuclei_tiles(B, 256, start=3))
    eng.infer_fixed_load_async(tiles, torch.from_numpy(rois).to(tiles.device), 40, hip.CH_SWAP)
    eng.check()
    counts = eng.buffer('roi_fallback_count').cpu().numpy()
    assert counts[0] > 512 and counts[1] > 1024, counts           # long lists: the one-workgroup-per-box / one-wave-per-RoI forms ran
    R = B * n
    feats_long = eng.buffer('bbox_feats')[:R].cpu().clone()
    x = [_nchw(eng.buffer(f'x{i}')[:B]) for i in range(4)]
    sem_feat = _nchw(eng.buffer('sem_feat')[:B])
    r2 = eng.buffer('rois_stage2')[:R].cpu()
    pick = torch.from_numpy(np.sort(rng.choice(R, 240, replace=False)))
    with torch.no_grad():
        ref = O.bbox_feats(x, sem_feat, r2[pick])
    got = feats_long[pick].reshape(len(pick), 7, 7, 64).permute(0, 3, 1, 2)
    err = float((got - ref).abs().max())
    print(f'long lists: big {int(counts[0])}, mid {int(counts[1])}; max |err| against the oracle on 240 RoIs {err:.2e}')
    assert err <= 2e-4, err
    # the same tiles with the first m RoIs of each alone: short lists
    eng.infer_fixed_load_async(tiles, torch.from_numpy(np.ascontiguousarray(rois[:, :m])).to(tiles.device), 40, hip.CH_SWAP)
    eng.check()
    c2 = eng.buffer('roi_fallback_count').cpu().numpy()
    assert 0 < c2[0] <= 512 and 0 < c2[1] <= 1024, c2
    feats_short = eng.buffer('bbox_feats')[:B * m].cpu()
    for b in range(B):
        assert torch.equal(feats_short[b * m:(b + 1) * m], feats_long[b * n:b * n + m]), f'tile {b}: features depend on the rest of the batch'


def test_attention_pool_fp16_switch_is_the_reference_on_cuda_arithmetic(hip_device):
    """nuhtc_config.att_pool_fp16 = 1: the level-2 / level-3 attention-pool tables as the reference computes them when its feature maps
    sit on a CUDA device (it casts that branch to fp16 there, nuhtc/models/roi_extractors_cus.py:203,231) against the oracle's fp16 mode,
    which runs the reference's own tensor expressions on fp16 tensors.  Every value is an fp16 number; it agrees with the oracle to one
    fp16 unit (the fp32 sums inside each rounded operation run in another order) and nearly all values agree bit for bit.  Everything
    upstream is untouched by the switch, and the default engine keeps the fp32 tables."""
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(3)
    B = 2
    e16 = Engine(sd, device=0, max_batch=B, tile=(256, 256), att_pool_fp16=1)
    e32 = Engine(sd, device=0, max_batch=B, tile=(256, 256))
    tiles = e16.to_device(synth.nuclei_tiles(B, 256, start=11))
    for e in (e16, e32):
        e.infer_async(tiles, hip.CH_SWAP)
        e.check()
    changed = 0
    for lvl, stride in ((2, 16), (3, 32)):
        x = _nchw(e16.buffer(f'x{lvl}')[:B])
        assert torch.equal(x, _nchw(e32.buffer(f'x{lvl}')[:B]))
        H, W = x.shape[2:]
        cy, cx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        one = np.stack([cx * stride + 1.0, cy * stride + 1.0, cx * stride + stride - 1.0, cy * stride + stride - 1.0], -1).reshape(-1, 4)
        rois = torch.from_numpy(np.concatenate([np.concatenate([np.full((H * W, 1), float(b)), one], 1) for b in range(B)]).astype(np.float32))
        with torch.no_grad():
            ref16 = O.attention_pool(x, rois, stride, fp16=True)
            ref32 = O.attention_pool(x, rois, stride, fp16=False)
        got16 = e16.buffer(f'G{lvl}')[:B].reshape(-1, 64).cpu()
        got32 = e32.buffer(f'G{lvl}')[:B].reshape(-1, 64).cpu()
        assert torch.equal(got16, got16.half().float())                      # fp16 numbers, carried in fp32
        unit = torch.maximum(ref16.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
        off = (got16 - ref16).abs()
        same = float((got16 == ref16).float().mean())
        print(f'level {lvl}: {got16.numel()} values, bit-equal to the oracle fp16 mode {same:.4f}, worst {float((off / unit).max()):.2f} fp16 units; '
              f'fp16 vs fp32 tables differ by up to {float((got16 - got32).abs().max()):.2e}')
        assert bool((off <= unit).all()) and same >= 0.97
        assert float((got32 - ref32).abs().max()) <= 2e-5
        changed += int((got16 != got32).sum())
    assert changed > 0
    n16, n32 = e16.counts[:B].cpu().numpy(), e32.counts[:B].cpu().numpy()
    print('detections per tile, fp16 switch / default:', n16.tolist(), n32.tolist())
    assert (np.abs(n16 - n32) <= np.maximum(3, n32 // 20)).all()
