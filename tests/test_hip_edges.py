"""GPU edge cases and full-size properties of the tile path (through the C ABI).

Edge cases the reference's own surface implies: tiles without any detection, constant / noise tiles, a batch of one, a
non-square tile, a detection cap that truncates, a score threshold nothing passes.  Full-size properties at BASELINE
configs[1] (B=16, 256x256), which the CPU oracle cannot cover in seconds: determinism, batch-permutation equivariance,
idempotence of the per-tile mask-NMS keep set, structural invariants of the outputs."""
import numpy as np
import pytest

import parity_util as P

pytestmark = pytest.mark.gpu


def _pad5(r, nc=5):
    return r


def _flat(res):
    b = np.concatenate(res[0], 0)
    m = [x for cl in res[1] for x in cl]
    return b, m


def test_blank_noise_and_single_tile_batches(hip_device):
    from nuhtc_amd import synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(3, obj_bias=0.0)
    rng = np.random.default_rng(5)
    tiles = np.stack([np.full((64, 64, 3), 255, np.uint8), np.zeros((64, 64, 3), np.uint8), synth.nuclei_tiles(1, 64, start=9)[0],
                      rng.integers(0, 256, (64, 64, 3), dtype=np.uint8), np.full((64, 64, 3), 127, np.uint8)])
    eng = Engine(sd, device=0, max_batch=8, tile=(64, 64))
    got = eng(tiles, 1)
    ref, it = O.Oracle(sd)(tiles, 1, keep=True)
    vals = P.oracle_paste_values(O, it, (64, 64))
    for i, (g, r) in enumerate(zip(got, ref)):
        rep, fails = P.compare_strict(r, g, values=vals[i])   # exact counts; a miss must sit on a threshold
        print(f'tile {i}: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        assert not fails, (i, fails)
    # a batch of one gives the same answer as the same tile inside a batch
    for i in (0, 2, 3):
        one = eng(tiles[i:i + 1], 1)[0]
        assert all(np.array_equal(a, b) for a, b in zip(one[0], got[i][0]))
        assert all(len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(one[1], got[i][1]))
    # contours / keep flags cope with tiles that have no detections
    B = eng.infer_async(eng.to_device(tiles), 1)
    eng.check()
    rings = eng.contours(B)
    counts = eng.counts[:B].cpu().numpy()
    keep = eng.keep[:B].cpu().numpy()
    for b in range(B):
        assert len(rings[b]) <= counts[b] and not keep[b, counts[b]:].any()


def test_nothing_passes_the_score_threshold(hip_device):
    from nuhtc_amd import synth, weights
    from nuhtc_amd.engine import Engine
    sd = weights.bench_state_dict(3, obj_bias=0.0)
    eng = Engine(sd, device=0, max_batch=4, tile=(64, 64), score_thr=0.9999)
    got = eng(synth.nuclei_tiles(3, 64, start=2), 1)
    for bbox, segm in got:
        assert all(b.shape == (0, 5) for b in bbox) and all(len(s) == 0 for s in segm)
    assert eng.counts[:3].cpu().numpy().tolist() == [0, 0, 0]
    assert all(len(r) == 0 for r in eng.contours(3))


def test_non_square_tile_and_detection_cap(hip_device):
    from nuhtc_amd import synth, weights
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    sd = weights.bench_state_dict(4, obj_bias=0.5)
    base = synth.nuclei_tiles(2, 128, start=21)
    tiles = np.ascontiguousarray(base[:, :64, :96])               # 64 x 96 tiles
    eng = Engine(sd, device=0, max_batch=2, tile=(64, 96))
    got = eng(tiles, 0)
    ref, it = O.Oracle(sd)(tiles, 0, keep=True)
    vals = P.oracle_paste_values(O, it, (64, 96))
    for i, (g, r) in enumerate(zip(got, ref)):
        rep, fails = P.compare_strict(r, g, values=vals[i])
        print(f'tile {i}: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        assert rep['n_got'] > 0 and not fails, fails
        assert all(m.shape == (64, 96) for cl in g[1] for m in cl)
    # max_per_img truncates to the best-scoring detections of the uncapped run
    cap = 7
    engc = Engine(sd, device=0, max_batch=2, tile=(64, 96), max_per_img=cap)
    gotc = engc(tiles, 0)
    for g, gc in zip(got, gotc):
        b, _ = _flat(g)
        bc, _ = _flat(gc)
        assert len(bc) == min(cap, len(b))
        top = b[np.argsort(-b[:, 4], kind='stable')[:len(bc)]]
        assert np.allclose(np.sort(bc[:, 4]), np.sort(top[:, 4]), atol=1e-6)


def test_full_size_properties_b16(hip_device):
    """BASELINE configs[1] size: B=16 tiles of 256x256."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    sd = weights.bench_state_dict(0)
    eng = Engine(sd, device=0, max_batch=16)
    tiles = synth.nuclei_tiles(16, 256, start=0)
    dev = eng.to_device(tiles)

    def snapshot():
        B = eng.infer_async(dev, hip.CH_SWAP)
        eng.check()
        return [t[:B].clone() for t in (eng.counts, eng.boxes, eng.labels, eng.masks, eng.areas, eng.keep)]
    a = snapshot()
    b = snapshot()
    counts = a[0].cpu().numpy()
    assert counts.min() > 0 and counts.max() <= 500
    # determinism: a second run is bit-identical in every output buffer (valid rows)
    for x, y in zip(a, b):
        for t in range(16):
            n = counts[t]
            assert torch.equal(x[t][:n] if x[t].dim() else x[t], y[t][:n] if y[t].dim() else y[t])
    # permutation equivariance: reversing the batch reverses the results
    perm = torch.arange(15, -1, -1, device=dev.device)
    B = eng.infer_async(dev[perm].contiguous(), hip.CH_SWAP)
    eng.check()
    c = [t[:B].clone() for t in (eng.counts, eng.boxes, eng.labels, eng.masks, eng.areas, eng.keep)]
    for x, y in zip(a, c):
        for t in range(16):
            n = counts[t]
            assert torch.equal(x[t][:n] if x[t].dim() else x[t], y[15 - t][:n] if y[15 - t].dim() else y[15 - t])
    # structural invariants
    boxes, labels, masks, areas, keep = (t.cpu().numpy() for t in a[1:])
    for t in range(16):
        n = counts[t]
        bx = boxes[t, :n]
        assert (bx[:, 4] >= 0.35).all() and (bx[:, 4] <= 1).all() and (np.diff(bx[:, 4]) <= 1e-7).all()     # NMS order: score descending
        assert (bx[:, 0] >= 0).all() and (bx[:, 1] >= 0).all() and (bx[:, 2] <= 256).all() and (bx[:, 3] <= 256).all()
        assert ((labels[t, :n] >= 0) & (labels[t, :n] < 5)).all()
        bits = np.unpackbits(masks[t, :n].view(np.uint8).reshape(n, 256, 32), axis=-1, bitorder='little')
        assert np.array_equal(bits.reshape(n, -1).sum(1), areas[t, :n])                                    # popcount areas
        # mask pixels lie inside the (integer-expanded) box
        ys, xs = np.nonzero(bits.any(0))
        k = np.nonzero(keep[t, :n])[0]
        # keep set: margin / min-area filter holds, and mask-NMS at 0.05 over the kept set suppresses nothing more
        assert (areas[t, k] >= 10).all()
        assert (bx[k, 0] >= 2).all() and (bx[k, 1] >= 2).all() and (bx[k, 2] <= 254).all() and (bx[k, 3] <= 254).all()
        if len(k) > 1:
            f = bits[k].reshape(len(k), -1).astype(np.float32)
            inter = f @ f.T
            ar = f.sum(1)
            iou = inter / (ar[:, None] + ar[None, :] - inter)
            np.fill_diagonal(iou, 0)
            assert iou.max() <= 0.05
        # and every dropped-but-eligible detection is overlapped by a better kept one (greedy NMS completeness)
        elig = np.array([j for j in range(n) if j not in set(k.tolist()) and areas[t, j] >= 10 and bx[j, 0] >= 2 and bx[j, 1] >= 2
                         and bx[j, 2] <= 254 and bx[j, 3] <= 254], dtype=int)
        for j in elig:
            fj = bits[j].reshape(-1).astype(np.float32)
            better = [q for q in k if bx[q, 4] >= bx[j, 4]]
            fk = bits[better].reshape(len(better), -1).astype(np.float32)
            inter = fk @ fj
            iou = inter / (fk.sum(1) + fj.sum() - inter)
            assert iou.max() > 0.05


def test_pipeline_matches_single_engine(hip_device):
    """Three batches in flight on their own engines / streams give bit-identical outputs to one engine run batch by batch,
    in submission order, including a ragged last batch."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.engine import Engine
    from nuhtc_amd.pipeline import EnginePipeline
    sd = weights.bench_state_dict(2, obj_bias=0.0)
    tiles = synth.nuclei_tiles(14, 64, start=11)
    eng = Engine(sd, device=0, max_batch=4, tile=(64, 64))
    ref = []
    for i in range(0, 14, 4):
        B = eng.infer_async(eng.to_device(tiles[i:i + 4]), hip.CH_SWAP)
        eng.check()
        ref.append([t[:B].clone().cpu() for t in (eng.counts, eng.boxes, eng.labels, eng.masks, eng.keep)])
    pipe = EnginePipeline(sd, device=0, depth=3, max_batch=4, tile=(64, 64))
    got = []

    def collect():
        e, B, stream, tag = pipe.collect()
        with torch.cuda.stream(stream):
            got.append((tag, [t[:B].clone().cpu() for t in (e.counts, e.boxes, e.labels, e.masks, e.keep)]))
    for i in range(0, 14, 4):
        if pipe.full():
            collect()
        pipe.submit(tiles[i:i + 4], hip.CH_SWAP, tag=i)
    while pipe.pending:
        collect()
    assert [t for t, _ in got] == [0, 4, 8, 12]
    for (_, g), r in zip(got, ref):
        counts = r[0].numpy()
        for x, y in zip(g, r):
            for b in range(len(counts)):
                n = counts[b]
                assert torch.equal(x[b][:n] if x[b].dim() else x[b], y[b][:n] if y[b].dim() else y[b])
    with pytest.raises(RuntimeError):
        for _ in range(4):
            pipe.submit(tiles[:4], hip.CH_SWAP)


def test_exported_views_survive_a_resubmit_before_unpacking(hip_device):
    """A pipeline slot holds two exported batches; the slide loop may resubmit a slot BEFORE it unpacks the batch the slot just
    delivered.  The views export_read() hands out must then still be that batch's: three pinned host buffers per
    engine (Engine.EXPORT_BUFFERS), not two.  One engine (depth 1 -> every submit goes to the same slot), batches A, B queued, A
    collected, C submitted and COMPLETED before A's views are read: A's records must equal A's records read the orderly way."""
    import torch
    from nuhtc_amd import hip, synth, weights
    from nuhtc_amd.pipeline import EnginePipeline
    sd = weights.bench_state_dict(2, obj_bias=0.0)
    batches = [synth.nuclei_tiles(4, 64, start=s) for s in (11, 31, 51, 71)]
    pipe = EnginePipeline(sd, device=0, depth=1, max_batch=4, tile=(64, 64))
    assert pipe.engines[0].EXPORT_BUFFERS >= pipe.per_slot + 1
    snap = lambda g: {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in g.items()}
    # orderly: every batch unpacked (copied) before the next submit
    want = []
    for t in batches:
        pipe.submit(t, hip.CH_SWAP, export=True)
        e, B, st, _ = pipe.collect()
        want.append(snap(e.export_read(pipe.last_turn)))
    assert all(w['n'] > 0 for w in want) and len({w['n'] for w in want}) > 1
    # disorderly: A, B queued; collect A -> views (NOT copied); submit C, collect B, submit D: both later copies have landed
    pipe.submit(batches[0], hip.CH_SWAP, export=True)
    pipe.submit(batches[1], hip.CH_SWAP, export=True)
    e, B, st, _ = pipe.collect()
    views_a = e.export_read(pipe.last_turn)
    pipe.submit(batches[2], hip.CH_SWAP, export=True)
    e, B, st, _ = pipe.collect()
    views_b = e.export_read(pipe.last_turn)
    torch.cuda.synchronize()            # C's copy has completed: with two buffers it would have landed in A's
    for got, ref in ((views_a, want[0]), (views_b, want[1])):
        assert got['n'] == ref['n']
        for k in ('tile', 'slot', 'boxes', 'labels', 'cn', 'crop_box', 'crop_area'):
            assert np.array_equal(got[k], ref[k]), k
        for k in range(ref['n']):          # (vertices past a contour's length are whatever an earlier batch left there)
            assert np.array_equal(got['xy'][k, :max(int(ref['cn'][k]), 0)], ref['xy'][k, :max(int(ref['cn'][k]), 0)]), k
        assert got['crop_total'] == ref['crop_total'] and np.array_equal(got['crop_words'][:ref['crop_total']], ref['crop_words'][:ref['crop_total']])
    e, B, st, _ = pipe.collect()
    assert e.export_read(pipe.last_turn)['n'] == want[2]['n']
    pipe.close()


def test_mag20_scale_factor_4(hip_device):
    """tools/infer_wsi.py:416-419 sets MultiScaleFlipAug.scale_factor = 80 / mag: a 20x slide runs at scale_factor 4 (a 64-px tile
    is a 256-px network input; cv2's x4 8-bit resize, boxes / masks scaled back by 4).  Engine vs oracle, strict."""
    from nuhtc_amd import synth, weights
    from nuhtc_amd.config import Config, engine_options, set_test_scale_factor
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    import os
    cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'))
    assert set_test_scale_factor(cfg, 20) == 4.0 and engine_options(cfg)['scale_factor'] == 4.0
    sd = weights.bench_state_dict(6, obj_bias=0.0)
    tiles = synth.nuclei_tiles(3, 64, start=50)
    eng = Engine(sd, device=0, max_batch=4, tile=(64, 64), scale_factor=4.0)
    got = eng(tiles, 1)
    img = eng.buffer('img')[:3].cpu().numpy()                       # (B, 256, 256, 3) network input
    ref_img = O.preprocess(tiles, 1, scale=4).permute(0, 2, 3, 1).numpy()
    assert img.shape == ref_img.shape == (3, 256, 256, 3) and np.abs(img - ref_img).max() <= 1e-6
    ref, it = O.Oracle(sd, scale=4.0)(tiles, 1, keep=True)
    vals = P.oracle_paste_values(O, it, (64, 64), 4.0)
    n = 0
    for i, (g, r) in enumerate(zip(got, ref)):
        rep, fails = P.compare_strict(r, g, values=vals[i])
        print(f'mag 20 tile {i}: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        assert not fails, (i, fails)
        n += rep['n_got']
    assert n > 0
    # a scale the engine cannot take is refused at creation, not silently rounded
    from nuhtc_amd.engine import HipError
    with pytest.raises(HipError):
        Engine(sd, device=0, max_batch=1, tile=(64, 64), scale_factor=2.7)


def test_image_size_not_a_multiple_of_32(hip_device):
    """A 72 x 90 image: the reference's test pipeline resizes it to 144 x 180 (img_shape), Pad(size_divisor=32) makes the network input
    160 x 192 (pad_shape, zeros after normalisation), RPN proposals / refined boxes / detections are clipped to img_shape, the
    component proposals are computed on the semantic logits interpolated to img_shape, masks are pasted into 72 x 90 (ori_shape).
    Golden `pad_b2` is the reference's own output for this case; engine vs oracle stage by stage and engine vs oracle / golden end
    to end with the strict gates.  Also through the public API (inference_detector on an ndarray of that size)."""
    import torch
    import golden_util as G
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    g = G.load('pad_b2')
    sd = G.seeded_sd(g)
    tiles = g['tiles']
    B, (h, w) = len(tiles), tiles.shape[1:3]
    assert (h, w) == (72, 90)
    mode = int(g['channel_mode'])
    eng = Engine(sd, device=0, max_batch=B, tile=(h, w))
    assert (eng.cfg.tile_h, eng.cfg.tile_w, eng.cfg.valid_h, eng.cfg.valid_w) == (72, 96, 72, 90)
    eng.infer_async(eng.to_device(tiles), mode)
    got = eng.results(B)
    msgs = []
    # a1: resize + normalise + zero pad
    img = eng.buffer('img')[:B].cpu().numpy()
    ref_img = O.preprocess(tiles, mode).permute(0, 2, 3, 1).numpy()
    assert img.shape == ref_img.shape == (B, 160, 192, 3)
    assert np.abs(img - ref_img).max() <= 1e-6 and (img[:, 144:] == 0).all() and (img[:, :, 180:] == 0).all()
    # dense stages on the padded tensor
    with torch.no_grad():
        x = O.fpn(sd, O.backbone(sd, torch.from_numpy(ref_img).permute(0, 3, 1, 2)))
        sem_pred, sem_feat = O.semantic_head(sd, x)
    for i in range(4):
        err = float((eng.buffer(f'x{i}')[:B].cpu().permute(0, 3, 1, 2) - x[i]).abs().max())
        print(f'x{i}: {err:.2e}')
        assert err <= 2e-4, (i, err)
    # a10-a12 / a14 from the engine's own maps: proposals live inside img_shape = 144 x 180
    nchw = lambda t: t.cpu().permute(0, 3, 1, 2).contiguous()
    rp = [nchw(eng.buffer(f'rpn{i}')[:B]) for i in range(4)]
    rpn_ref = O.rpn_proposals([r[:, 0:3] for r in rp], [r[:, 3:15] for r in rp], (144, 180))
    rpn_counts = eng.buffer('rpn_counts')[:B].cpu().numpy()
    rpn = eng.buffer('rpn_props')[:B].cpu().numpy()
    cc_ref = O.cc_proposals(eng.buffer('sem_pred')[:B].cpu()[:, None], (144, 180))
    cc_counts = eng.buffer('cc_counts')[:B].cpu().numpy()
    cc = eng.buffer('cc_props')[:B].cpu().numpy()
    for i in range(B):
        a, b = rpn_ref[i].numpy(), rpn[i, :rpn_counts[i]]
        assert len(a) == len(b), (i, len(a), len(b))
        assert b[:, 2].max() <= 180.0 and b[:, 3].max() <= 144.0 and b[:, :4].max() > 150.0
        # in list order (both are in NMS order); the canonical sort is the fallback for exact score ties -- alone it mis-pairs rows whenever two
        # scores one ulp apart swap places between the CPU's and the GPU's sigmoid
        d = min(np.abs(a - b).max(), np.abs(G.canon_rows(a) - G.canon_rows(b)).max())
        print(f'tile {i}: {len(b)} rpn proposals, max diff {d:.2e}; cc proposals oracle/hip {len(cc_ref[i])}/{cc_counts[i]}')
        assert d <= 1e-3
        assert np.array_equal(cc_ref[i].numpy()[:, :4], cc[i, :cc_counts[i]]), i
    # end to end
    ref, it = O.Oracle(sd)(tiles, mode, keep=True)
    vals = P.oracle_paste_values(O, it, (h, w))
    for i in range(B):
        gd, gl = g[f'det{i}'], g[f'lab{i}']
        gm = np.unpackbits(g[f'masks{i}'], axis=-1).astype(bool)[..., :w]
        gold = ([gd[gl == c] for c in range(5)], [[gm[j] for j in range(len(gd)) if gl[j] == c] for c in range(5)])
        assert all(m.shape == (h, w) for cl in got[i][1] for m in cl)
        for tag, r, v in (('oracle', ref[i], vals[i]), ('golden', gold, None)):
            rep, fails = P.compare_strict(r, got[i], values=v, values_side='ref')
            print(f'pad_b2 tile {i} end-to-end vs {tag}: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
            msgs += [f'tile {i} vs {tag}: {f}' for f in fails]
        assert sum(len(b) for b in got[i][0]) > 100
    assert not msgs, '\n'.join(msgs)
    # the public API on an array of that size (inference_detector pads nothing itself: the engine does what Pad does)
    eng.close()
    import os
    import warnings
    from nuhtc_amd import apis
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = apis.init_detector(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'),
                                   None, 'cuda:0', max_batch=2)
    model.state_dict = sd
    bbox_res, segm_res = apis.inference_detector(model, tiles[0])
    n = sum(len(b) for b in bbox_res)
    assert n > 50 and all(m.shape == (h, w) and m.dtype == bool for cl in segm_res for m in cl)
    allb = np.concatenate(bbox_res, 0)
    assert allb[:, 2].max() <= w and allb[:, 3].max() <= h


def test_tile_size_without_window_padding_in_any_stage(hip_device):
    """112 x 112 tiles: the network input is 224 x 224, the four Swin stages have 56 / 28 / 14 / 7 tokens per side -- multiples of the 7 x 7
    window, so no stage pads (swin.py:341-343 pads nothing), the window image has no padding rows, the engine's padding-row list is empty and
    every padding mask of the attention kernel is zero; stage 4 is ONE window and its shifted block has no shift (7 <= window size is not the
    case mmdet special-cases: shift stays 3, the mask applies).  Engine vs oracle: FPN maps, then end to end with the strict gates."""
    import torch
    import golden_util as G
    from nuhtc_amd.engine import Engine
    from oracle import model as O
    g = G.load('five_b2')
    sd = G.seeded_sd(g)
    tiles = np.ascontiguousarray(g['tiles'][:, 40:152, 72:184])
    B, (h, w) = len(tiles), tiles.shape[1:3]
    assert (h, w) == (112, 112)
    mode = int(g['channel_mode'])
    eng = Engine(sd, device=0, max_batch=B, tile=(h, w))
    eng.infer_async(eng.to_device(tiles), mode)
    got = eng.results(B)
    ref_img = O.preprocess(tiles, mode)
    assert tuple(ref_img.shape) == (B, 3, 224, 224)
    with torch.no_grad():
        c = O.backbone(sd, ref_img)
        x = O.fpn(sd, c)
    assert [t.shape[-1] for t in c] == [56, 28, 14, 7]
    for i in range(4):
        ec = float((eng.buffer(f'c{i}')[:B].cpu().permute(0, 3, 1, 2) - c[i]).abs().max())
        ex = float((eng.buffer(f'x{i}')[:B].cpu().permute(0, 3, 1, 2) - x[i]).abs().max())
        print(f'c{i}: {ec:.2e}  x{i}: {ex:.2e}')
        assert ec <= 2e-4 + 2e-4 * float(c[i].abs().max()) and ex <= 2e-4, (i, ec, ex)
    ref, it = O.Oracle(sd)(tiles, mode, keep=True)
    vals = P.oracle_paste_values(O, it, (h, w))
    msgs = []
    for i in range(B):
        rep, fails = P.compare_strict(ref[i], got[i], values=vals[i], values_side='ref')
        print(f'112 px tile {i} end-to-end vs oracle: {P.fmt(rep)}', *rep['explained'], sep='\n    ')
        msgs += [f'tile {i}: {f}' for f in fails]
        assert sum(len(b) for b in got[i][0]) > 10
    assert not msgs, '\n'.join(msgs)
    eng.close()


def test_engine_creation_puts_the_submitting_thread_on_the_gpus_numa_node(hip_device, monkeypatch):
    """Engine(bind_host=True) calls nuhtc_bind_host_thread (the command processor reads every dispatch packet from host
    memory the submitter wrote; from the other socket that costs 0.3-0.4 ms per step): afterwards the calling thread runs on the CPUs sysfs
    lists as local to the GPU, and gets its mask back when the LAST engine that placed it is closed.  The default leaves the caller's mask
    alone (a library does not change process state unasked; NUHTC_HOST_AFFINITY=1 switches the default); results do not depend on it."""
    import os
    import torch
    from nuhtc_amd import synth, weights
    from nuhtc_amd.apis import init_detector
    from nuhtc_amd.engine import Engine
    pr = torch.cuda.get_device_properties(0)
    bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
    try:
        text = open(f'/sys/bus/pci/devices/{bdf}/local_cpulist').read().strip()
    except OSError:
        text = ''
    local = set()
    for part in filter(None, text.split(',')):
        a, _, b = part.partition('-')
        local |= set(range(int(a), int(b or a) + 1))
    before = os.sched_getaffinity(0)
    sd = weights.seeded_state_dict(0)
    tiles = synth.nuclei_tiles(2, 64)
    try:
        monkeypatch.delenv('NUHTC_HOST_AFFINITY', raising=False)
        e0 = Engine(sd, device=0, max_batch=2, tile=(64, 64))
        assert os.sched_getaffinity(0) == before                  # the default: untouched
        r0 = e0(tiles)
        e1 = Engine(sd, device=0, max_batch=2, tile=(64, 64), bind_host=True)
        e2 = Engine(sd, device=0, max_batch=2, tile=(64, 64), bind_host=True)
        now = os.sched_getaffinity(0)
        placed = bool(local and (local & before) and (local & before) != before)
        if local and (local & before):
            assert now == (local & before), (sorted(now)[:4], len(now), text)
        else:
            assert now == before           # no NUMA information for the device, or the caller's mask excludes the node
        r1 = e1(tiles)
        for (b0, m0), (b1, m1) in zip(r0, r1):
            assert all(np.array_equal(x, y) for x, y in zip(b0, b1))
            assert all(len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y)) for x, y in zip(m0, m1))
        e1.close()
        assert os.sched_getaffinity(0) == now                     # e2 still needs the placement
        e2.close()
        assert os.sched_getaffinity(0) == before                  # the last one gives the mask back
        monkeypatch.setenv('NUHTC_HOST_AFFINITY', '1')            # the environment switches the default of the argument
        e3 = Engine(sd, device=0, max_batch=2, tile=(64, 64))
        assert os.sched_getaffinity(0) == now
        e3.close()
        assert os.sched_getaffinity(0) == before
        monkeypatch.delenv('NUHTC_HOST_AFFINITY')
        det = init_detector(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'),
                            None, device='cuda:0', max_batch=2, bind_host=True)
        eng = det.engine((64, 64))                                # surfaced through init_detector
        assert os.sched_getaffinity(0) == now
        eng.close()
        assert os.sched_getaffinity(0) == before
        assert placed or now == before
    finally:
        os.sched_setaffinity(0, before)
