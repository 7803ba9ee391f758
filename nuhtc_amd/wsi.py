"""Host side of the whole-slide path (tools/infer_wsi.py:460-531 + tools/nuclei_merge.py:62-174).

Tile contract of the reference (`Whole_Slide_Bag_FP`, tools/wsi_core/WholeSlideImage.py:832-898): RGB uint8
(patch, patch, 3) tiles with their level-0 (x, y) origin; grid = np.arange(start, stop, step) (:460-466).
OpenSlide / h5py are not available offline, so tiles come from arrays (`tile_grid` over an in-memory image, or a
.npz with `tiles` + `coords`)."""
import numpy as np

from . import hip


def tile_grid(image, patch_size=256, step_size=192):
    """Grid tiling with zero padding past the edge (use_padding=True, WholeSlideImage.py:419-421,463-464)."""
    H, W = image.shape[:2]
    xs = np.arange(0, W, step_size)
    ys = np.arange(0, H, step_size)
    tiles, coords = [], []
    for y in ys:
        for x in xs:
            t = np.zeros((patch_size, patch_size, 3), np.uint8)
            sub = image[y:y + patch_size, x:x + patch_size]
            t[:sub.shape[0], :sub.shape[1]] = sub
            tiles.append(t)
            coords.append((x, y))
    return np.stack(tiles), np.array(coords, np.int64)


def _unpack(eng, B, i0, coords, P, rec):
    """Device outputs of one finished batch -> detection records (kept detections only), in slide coordinates."""
    rings = eng.contours(B)             # slot -> open contour in tile pixels, traced on the GPU
    counts = eng.counts[:B].cpu().numpy()
    boxes = eng.boxes[:B].cpu().numpy()
    labels = eng.labels[:B].cpu().numpy()
    keep = eng.keep[:B].cpu().numpy()
    for b in range(B):
        n = int(counts[b])
        idx = np.nonzero(keep[b, :n])[0]
        if len(idx) == 0:
            continue
        # class-major order like np.concatenate(result[0]) in the reference, then score order from mask_nms
        order = idx[np.lexsort((idx, labels[b, idx]))]
        order = order[np.argsort(boxes[b, order, 4], kind='stable')[::-1]]
        words = eng.masks[b, torch_index(order, eng)].cpu().numpy().view(np.uint32)
        bits = np.unpackbits(words.view(np.uint8).reshape(len(order), P, P // 8), axis=-1, bitorder='little').astype(bool)
        ox, oy = int(coords[i0 + b][0]), int(coords[i0 + b][1])
        for k, j in enumerate(order):
            # a pasted mask lives inside the integer hull of its box (fcn_mask_head.py:344-412): search only there
            bx = boxes[b, j]
            hy0, hy1 = max(int(np.floor(bx[1])) - 1, 0), min(int(np.ceil(bx[3])) + 1, P)
            hx0, hx1 = max(int(np.floor(bx[0])) - 1, 0), min(int(np.ceil(bx[2])) + 1, P)
            crop = bits[k, hy0:hy1, hx0:hx1]
            rows, cols = np.flatnonzero(crop.any(1)), np.flatnonzero(crop.any(0))
            if len(rows) == 0:
                continue
            y0, y1, x0, x1 = hy0 + rows[0], hy0 + rows[-1] + 1, hx0 + cols[0], hx0 + cols[-1] + 1
            rec['tile'].append(i0 + b)
            rec['box'].append(boxes[b, j, :4].astype(np.float64) + np.array([ox, oy, ox, oy]))
            rec['score'].append(float(boxes[b, j, 4]))
            rec['label'].append(int(labels[b, j]))
            rec['mask'].append((bits[k, y0:y1, x0:x1].copy(), ox + int(x0), oy + int(y0)))
            c = rings[b][int(j)]
            rec['ring'].append(np.concatenate([c, c[:1]], 0) + np.array([ox, oy], np.int64))   # mask2inst + contour_map


def torch_index(order, eng):
    import torch
    return torch.from_numpy(np.ascontiguousarray(order)).to(eng.device)


def infer_tiles(model, tiles, coords, batch_size=16, depth=3):
    """Run the engine over `tiles` (N,P,P,3) and return per-detection records that survive the per-tile margin /
    min-area filter + mask-NMS (computed on the GPU, tools/infer_wsi.py:510-531), in slide coordinates.  `depth` batches
    are kept in flight (nuhtc_amd.pipeline): the host unpacks batch i while the GPU runs batch i+1.

    Returns dict(tile, box (n,4) float64 slide px, score, label, mask (list of (bool crop, x0, y0)), ring (closed
    (n+1,2) int64 contour in slide px, traced on the GPU: nuhtc_mask_contours))."""
    import torch
    P = tiles.shape[1]
    rec = dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
    pipe = model.pipeline(tiles.shape[1:3], depth)

    def finish():
        eng, B, stream, i0 = pipe.collect()
        with torch.cuda.stream(stream):
            _unpack(eng, B, i0, coords, P, rec)
            stream.synchronize()

    for i in range(0, len(tiles), batch_size):
        if pipe.full():
            finish()
        pipe.submit(tiles[i:i + batch_size], hip.CH_SWAP, tag=i)
    while pipe.pending:
        finish()
    return rec


def pack_masks(masks):
    """[(bool crop, x0, y0)] -> (boxes int32 [n,4] (x1,y1 exclusive), areas int32 [n], bits uint32 [...], bit_off int64 [n]):
    the crop layout nuhtc_merge_overlap takes (rows of (w+31)//32 words, pixel x in bit x&31 of word x>>5)."""
    n = len(masks)
    boxes = np.zeros((n, 4), np.int32)
    areas = np.zeros(n, np.int32)
    off = np.zeros(n, np.int64)
    parts, cur = [], 0
    for i, (m, x0, y0) in enumerate(masks):
        h, w = m.shape
        boxes[i] = (x0, y0, x0 + w, y0 + h)
        areas[i] = int(m.sum())
        off[i] = cur
        if h and w:
            wpr = (w + 31) // 32
            row = np.zeros((h, wpr * 32), np.uint8)
            row[:, :w] = m
            words = np.packbits(row.reshape(h, wpr, 4, 8), axis=-1, bitorder='little').reshape(h, wpr, 4).view(np.uint32).reshape(-1)
            parts.append(words)
            cur += words.size
    bits = np.concatenate(parts) if parts else np.zeros(1, np.uint32)
    return boxes, areas, bits, off


def merge_overlap_packed(boxes, scores, areas, bits, off, overlap_threshold=0.05, device=0):
    """nuhtc_merge_overlap on packed crops (see pack_masks); numpy in, kept indices (ascending) out."""
    import ctypes
    import torch
    n = len(scores)
    if n == 0:
        return np.zeros(0, np.int64)
    lib = hip.load()
    dev = torch.device('cuda', device)
    arrs = (np.ascontiguousarray(boxes, np.int32), np.ascontiguousarray(scores, np.float32), np.ascontiguousarray(areas, np.int32),
            np.ascontiguousarray(bits, np.uint32).view(np.int32), np.ascontiguousarray(off, np.int64))
    t = [torch.from_numpy(a).to(dev) for a in arrs]
    keep = torch.zeros(n, dtype=torch.uint8, device=dev)
    vp = lambda x: ctypes.c_void_p(x.data_ptr())
    b = arrs[0]
    with torch.cuda.device(dev):
        rc = lib.nuhtc_merge_overlap(device, vp(t[0]), vp(t[1]), vp(t[2]), vp(t[3]), vp(t[4]), n, float(overlap_threshold),
                                     int(b[:, 0].min()), int(b[:, 1].min()), int(b[:, 2].max()), int(b[:, 3].max()),
                                     vp(keep), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc:
        raise RuntimeError(f'nuhtc_merge_overlap failed ({rc})')
    return np.nonzero(keep.cpu().numpy())[0].astype(np.int64)


def merge_overlap(rec, overlap_threshold=0.05, device=0):
    """Cross-tile duplicate removal, strategy 'probability' of tools/nuclei_merge.py:62-174: detections sorted by
    score (descending), each still-alive one suppresses every later one it overlaps with IoU > threshold.
    The reference intersects shapely polygons of the cv2 contours; here IoU is taken on the pixel masks the
    polygons are traced from (same objects, pixel-area instead of polygon-area IoU).  Runs on the GPU
    (nuhtc_merge_overlap, csrc/merge.hip).  Returns kept indices (ascending)."""
    if len(rec['score']) == 0:
        return np.zeros(0, np.int64)
    boxes, areas, bits, off = pack_masks(rec['mask'])
    return merge_overlap_packed(boxes, rec['score'], areas, bits, off, overlap_threshold, device)
