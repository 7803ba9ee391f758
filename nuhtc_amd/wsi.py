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


def _gather_sync(eng, B):
    """Synchronous twin of Engine.export_async / export_read (any number of kept detections)."""
    import torch
    K = eng.cfg.max_per_img
    eng.contours_async(B)
    counts = eng.counts[:B].cpu().numpy()
    keep = eng.keep[:B].cpu().numpy()
    kept = (keep != 0) & (np.arange(K)[None, :] < counts[:, None])
    tile, slot = np.nonzero(kept)
    sel = torch.from_numpy(np.stack([tile, slot])).to(eng.device)
    return dict(n=len(tile), tile=tile, slot=slot, boxes=eng.boxes[sel[0], sel[1]].cpu().numpy(), labels=eng.labels[sel[0], sel[1]].cpu().numpy(),
                cn=eng.contour_n[sel[0], sel[1]].cpu().numpy(), xy=eng.contour_xy[sel[0], sel[1]].cpu().numpy(),
                words=eng.masks[sel[0], sel[1]].reshape(len(tile), -1).cpu().numpy().view(np.uint32))


class PackedMasks:
    """Mask crops of a slide's detections in the packed layout the device produces and nuhtc_merge_overlap takes: `boxes` int32 (n, 4)
    crop rectangles in slide pixels (x1, y1 exclusive), `areas` int32 (n,), `bits` uint32 words (rows of (w + 31) // 32 words, crop
    column x in bit x & 31 of word x >> 5), `off` int64 (n,) word offset of each crop.  Behaves like the list of
    (bool crop, x0, y0) the per-detection path builds: len(), indexing and iteration decode crops on demand."""

    def __init__(self, boxes, areas, bits, off):
        self.boxes, self.areas, self.bits, self.off = boxes, areas, bits, off
        self.arrays = None      # set by infer_tiles: the records' other fields as arrays (pack_records' fast path)

    def __len__(self):
        return len(self.areas)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        x0, y0, x1, y1 = (int(v) for v in self.boxes[i])
        h, w = y1 - y0, x1 - x0
        wpr = (w + 31) // 32
        words = self.bits[int(self.off[i]):int(self.off[i]) + h * wpr].reshape(h, wpr)
        return np.unpackbits(words.view(np.uint8), axis=-1, bitorder='little')[:, :w].astype(bool), x0, y0

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __add__(self, other):               # like the list it stands in for: concatenation gives a plain list of decoded crops
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def subset(self, keep):
        """The crops `keep` (indices) re-packed contiguously."""
        keep = np.asarray(keep, np.int64)
        b = self.boxes[keep]
        sizes = ((b[:, 3] - b[:, 1]).astype(np.int64) * ((b[:, 2] - b[:, 0] + 31) // 32)) if len(keep) else np.zeros(0, np.int64)
        new_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(keep) else np.zeros(0, np.int64)
        idx = np.arange(int(sizes.sum()), dtype=np.int64) + np.repeat(self.off[keep] - new_off, sizes)
        return PackedMasks(b, self.areas[keep], self.bits[idx] if len(idx) else np.zeros(0, np.uint32), new_off)


class RaggedRings:
    """The closed rings of a slide's records as ONE vertex array: ring i = flat[off[i]:off[i + 1]] ((n_i, 2) int64, a view).  Stands in for
    the list of per-record arrays (len(), indexing, iteration); a padded (records, longest ring, 2) array of a 10 000-tile slide was
    a quarter of a gigabyte that the loop's last step had to assemble while the GPU idled."""

    def __init__(self, flat, ring_n):
        self.flat = flat
        self.n = np.asarray(ring_n, np.int64)
        self.off = np.concatenate([[0], np.cumsum(self.n)]).astype(np.int64)

    def __len__(self):
        return len(self.n)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        return self.flat[self.off[i]:self.off[i + 1]]

    def __iter__(self):
        return (self.flat[self.off[i]:self.off[i + 1]] for i in range(len(self)))

    def __add__(self, other):               # like the list it stands in for: concatenation gives a plain list of rings
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def take(self, keep):
        """-> (vertices of the rings `keep`, concatenated; their lengths)."""
        keep = np.asarray(keep, np.int64)
        n = self.n[keep]
        if len(keep) == len(self) and (len(keep) == 0 or (keep == np.arange(len(self))).all()):
            return self.flat, n
        new_off = np.cumsum(n) - n
        idx = np.arange(int(n.sum()), dtype=np.int64) + np.repeat(self.off[keep] - new_off, n)
        return self.flat[idx], n


def _unpack_packed(eng, g, i0, coords, parts):
    """Vectorised twin of _unpack for a batch exported with device crops (Engine.export_async -> nuhtc_export_crops): appends one
    dict of arrays (the batch's records in the order _unpack produces) to `parts`."""
    from . import contours as host
    n = g['n']
    tile, slot, boxes, labels = g['tile'], g['slot'], g['boxes'], g['labels']
    cls_major = np.lexsort((slot, labels, tile))
    order = cls_major[np.lexsort((np.arange(n), boxes[cls_major, 4], -tile[cls_major]))[::-1]]
    cb = g['crop_box'][order].astype(np.int64)
    order = order[cb[:, 2] > cb[:, 0]]                      # a mask without a set pixel is no record (tools/infer_wsi.py never sees one)
    if len(order) == 0:
        return
    cb = g['crop_box'][order].astype(np.int64)
    org = np.asarray(coords)[i0 + tile[order]].astype(np.int64)          # (m, 2) tile origins (x, y)
    org4 = np.concatenate([org, org], 1)
    sizes = (cb[:, 3] - cb[:, 1]) * ((cb[:, 2] - cb[:, 0] + 31) // 32)
    src = g['crop_off'][order].astype(np.int64)
    fits = src + sizes <= g['pool']
    if fits.all():
        idx = np.arange(int(sizes.sum()), dtype=np.int64) + np.repeat(src - (np.cumsum(sizes) - sizes), sizes)
        bits = g['crop_words'][idx]
    else:                                                                 # crops past the pool: cut from the full masks on the device
        chunks = []
        for k, ok in zip(order, fits):
            if ok:
                chunks.append(g['crop_words'][int(g['crop_off'][k]):int(g['crop_off'][k]) + int(sizes[len(chunks)])])
            else:
                x0, y0, x1, y1 = g['crop_box'][k]
                m = eng.export_full_mask(int(k))[y0:y1, x0:x1]
                chunks.append(pack_masks([(m, 0, 0)])[2])
        bits = np.concatenate(chunks)
    # closed rings in slide coordinates, ragged: ring k has cn[k] traced vertices + its first vertex again (mask2inst closes the ring,
    # tools/infer_wsi.py:51-58); one gather from the export's (records, capacity, 2) vertex block
    cn = g['cn'][order].astype(np.int64)
    hosted = {}
    for j in np.flatnonzero(cn <= 0):                                     # contour over the device capacities: host mirror on the crop
        x0, y0, x1, y1 = cb[j]
        wpr = (x1 - x0 + 31) // 32
        o = int(np.cumsum(sizes)[j] - sizes[j])
        m = np.unpackbits(bits[o:o + sizes[j]].reshape(y1 - y0, wpr).view(np.uint8), axis=-1, bitorder='little')[:, :x1 - x0].astype(bool)
        hosted[int(j)] = host.trace_outer_contour(m) + np.array([x0, y0], np.int64)
        cn[j] = len(hosted[int(j)])
    ring_n = cn + 1
    off = np.cumsum(ring_n) - ring_n
    total = int(ring_n.sum())
    rec_of = np.repeat(np.arange(len(order)), ring_n)
    pos = np.arange(total, dtype=np.int64) - np.repeat(off, ring_n)
    pos[pos == np.repeat(cn, ring_n)] = 0                                 # the closing vertex is the first one
    dev_ok = np.ones(len(order), bool)
    dev_ok[list(hosted)] = False
    sel = dev_ok[rec_of]
    flat = np.zeros((total, 2), np.int64)
    flat[sel] = g['xy'][order[rec_of[sel]], pos[sel]]
    for j, c in hosted.items():
        flat[off[j]:off[j] + cn[j]] = c
        flat[off[j] + cn[j]] = c[0]
    flat += org[rec_of]
    parts.append(dict(tile=i0 + tile[order], box=boxes[order, :4].astype(np.float64) + org4, score=boxes[order, 4].astype(np.float64),
                      label=labels[order].astype(np.int64), crop_box=(cb + org4).astype(np.int32), area=g['crop_area'][order].astype(np.int32),
                      bits=bits, sizes=sizes, ring_n=ring_n, ring_flat=flat))


def _records_from_parts(parts):
    """Per-batch arrays of _unpack_packed -> the record dict of infer_tiles (lists for the scalar fields and the rings, PackedMasks
    for the mask crops, which also carries the scalar fields as whole-slide arrays for the vectorised consumers)."""
    if not parts:
        return dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
    cat = lambda k: np.concatenate([p[k] for p in parts], 0)
    tile, box, score, label, ring_n = cat('tile'), cat('box'), cat('score'), cat('label'), cat('ring_n')
    sizes = cat('sizes')
    off = (np.cumsum(sizes) - sizes).astype(np.int64)
    masks = PackedMasks(cat('crop_box'), cat('area'), cat('bits'), off)
    rings = RaggedRings(cat('ring_flat'), ring_n)
    rec = dict(tile=tile.tolist(), box=list(box), score=score.tolist(), label=label.tolist(), mask=masks, ring=rings)
    masks.arrays = dict(tile=tile, box=box, score=score, label=label, rings=rings)   # the scalar fields as whole-slide arrays
    return rec


def _unpack(eng, B, i0, coords, P, rec, exported=False):
    """Kept detections of one finished batch -> records in slide coordinates.  `exported`: the batch was submitted with
    export=True (its results already sit in the engine's pinned buffers); otherwise they are fetched here."""
    from . import contours as host
    g = eng.export_read() if exported else None
    if g is None:
        g = _gather_sync(eng, B)
    n = g['n']
    if n == 0:
        return
    if 'crop_box' in g:                     # exported with device crops: the array path, appended in list form
        parts = []
        _unpack_packed(eng, g, i0, coords, parts)
        if parts:
            _extend(rec, _records_from_parts(parts))
        return
    tile, slot, boxes, labels = g['tile'], g['slot'], g['boxes'], g['labels']
    # per tile: class-major order like np.concatenate(result[0]) in the reference, then score order from mask_nms
    # (stable descending sort by score of the class-major list); tiles ascending
    cls_major = np.lexsort((slot, labels, tile))
    order = cls_major[np.lexsort((np.arange(n), boxes[cls_major, 4], -tile[cls_major]))[::-1]]
    cfg = getattr(eng, 'cfg', None)                 # mask buffers: tile_h rows of tile_w / 32 words (tile_w = width padded to 32)
    P, PW = (cfg.tile_h, cfg.tile_w) if cfg is not None else (P, P)
    W = PW // 32
    for k in order:
        b, bx = int(tile[k]), boxes[k]
        # a pasted mask lives inside the integer hull of its box (fcn_mask_head.py:344-412): unpack only those rows
        hy0, hy1 = max(int(np.floor(bx[1])) - 1, 0), min(int(np.ceil(bx[3])) + 1, P)
        hx0, hx1 = max(int(np.floor(bx[0])) - 1, 0), min(int(np.ceil(bx[2])) + 1, PW)
        if hy1 <= hy0 or hx1 <= hx0:
            continue
        rows_w = g['words'][k].reshape(P, W)[hy0:hy1]
        crop = np.unpackbits(rows_w.view(np.uint8), axis=-1, bitorder='little')[:, hx0:hx1].astype(bool)
        rows, cols = np.flatnonzero(crop.any(1)), np.flatnonzero(crop.any(0))
        if len(rows) == 0:
            continue
        ox, oy = int(coords[i0 + b][0]), int(coords[i0 + b][1])
        y0, y1, x0, x1 = hy0 + rows[0], hy0 + rows[-1] + 1, hx0 + cols[0], hx0 + cols[-1] + 1
        rec['tile'].append(i0 + b)
        rec['box'].append(bx[:4].astype(np.float64) + np.array([ox, oy, ox, oy]))
        rec['score'].append(float(bx[4]))
        rec['label'].append(int(labels[k]))
        rec['mask'].append((crop[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1].copy(), ox + int(x0), oy + int(y0)))
        nv = int(g['cn'][k])
        if nv > 0:
            c = g['xy'][k, :nv].astype(np.int64)
        else:   # the contour overflowed the device capacities: host mirror on the full mask
            full = np.unpackbits(g['words'][k].reshape(P, W).view(np.uint8), axis=-1, bitorder='little').astype(bool)
            c = host.trace_outer_contour(full)
        rec['ring'].append(np.concatenate([c, c[:1]], 0) + np.array([ox, oy], np.int64))   # mask2inst + contour_map


def infer_tiles(model, tiles, coords, batch_size=16, depth=4):
    """Run the engine over `tiles` (N,P,P,3) and return per-detection records that survive the per-tile margin /
    min-area filter + mask-NMS (computed on the GPU, tools/infer_wsi.py:510-531), in slide coordinates.  `depth` engines
    are kept busy with up to two batches each (nuhtc_amd.pipeline): the host unpacks batch i while the GPU runs the next ones.

    Returns dict(tile, box (n,4) float64 slide px, score, label, mask (list of (bool crop, x0, y0)), ring (closed
    (n+1,2) int64 contour in slide px, traced on the GPU: nuhtc_mask_contours))."""
    import torch
    P = tiles.shape[1]
    parts = {}                 # first tile of the batch -> the batch's records (array form), joined in batch order at the end
    redo = []
    pipe = model.pipeline(tiles.shape[1:3], depth)

    def finish():
        eng, B, stream, i0 = pipe.collect()
        g = eng.export_read(pipe.last_turn)
        if g is not None and g['crop_total'] <= g['pool']:
            if g['n']:
                one = []
                _unpack_packed(eng, g, i0, coords, one)            # device crops: whole-batch array operations
                if one:
                    parts[i0] = one[0]
        else:
            # this batch alone held more kept detections (or mask-crop words) than the export buffers: its records have to come from
            # the engine's own tensors, which may belong to the slot's next batch by now -- it is run again, alone, at the end
            redo.append(i0)

    for i in range(0, len(tiles), batch_size):
        if pipe.full(export=True):
            finish()
        pipe.submit(tiles[i:i + batch_size], hip.CH_SWAP, tag=i, export=True)
    while pipe.pending:
        finish()
    for i0 in redo:            # the per-detection path (later batches kept the packed path)
        eng = pipe.submit(tiles[i0:i0 + batch_size], hip.CH_SWAP, tag=i0, export=False)
        eng, B, stream, _ = pipe.collect()
        with torch.cuda.stream(stream):
            one = dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
            _unpack(eng, B, i0, coords, P, one, exported=False)
            if one['tile']:
                parts[i0] = _part_from_lists(one)
    return _records_from_parts([parts[k] for k in sorted(parts)])


def _part_from_lists(rec):
    """List-form records of one batch (the per-detection path) -> the array form _unpack_packed produces."""
    n = len(rec['tile'])
    cb, area, bits, off = pack_masks(rec['mask'])
    sizes = (cb[:, 3] - cb[:, 1]).astype(np.int64) * ((cb[:, 2] - cb[:, 0] + 31) // 32)
    ring_n = np.array([len(r) for r in rec['ring']], np.int64)
    ring_flat = np.concatenate([np.asarray(r, np.int64).reshape(-1, 2) for r in rec['ring']], 0) if n else np.zeros((0, 2), np.int64)
    return dict(tile=np.asarray(rec['tile'], np.int64), box=np.stack(rec['box']).astype(np.float64), score=np.asarray(rec['score'], np.float64),
                label=np.asarray(rec['label'], np.int64), crop_box=cb.astype(np.int32), area=area.astype(np.int32),
                bits=bits[:int(sizes.sum())], sizes=sizes, ring_n=ring_n, ring_flat=ring_flat)


def _extend(rec, more):
    """Append the records `more` to the list-form record dict `rec` (mask crops decoded)."""
    for k in ('tile', 'box', 'score', 'label', 'ring'):
        rec[k].extend(list(more[k]))
    rec['mask'].extend(list(more['mask']))


def pack_masks(masks):
    """[(bool crop, x0, y0)] -> (boxes int32 [n,4] (x1,y1 exclusive), areas int32 [n], bits uint32 [...], bit_off int64 [n]):
    the crop layout nuhtc_merge_overlap takes (rows of (w+31)//32 words, pixel x in bit x&31 of word x>>5)."""
    if isinstance(masks, PackedMasks):
        return masks.boxes, masks.areas, (masks.bits if len(masks.bits) else np.zeros(1, np.uint32)), masks.off
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


def merge_overlap_packed(boxes, scores, areas, bits, off, overlap_threshold=0.05, device=0, overlap='polygon'):
    """nuhtc_merge_overlap on packed crops (see pack_masks); numpy in, kept indices (ascending) out.
    overlap: 'polygon' (the reference's measure: IoU of the ring polygons, exact) or 'mask' (IoU of the pixel sets)."""
    import ctypes
    import torch
    n = len(scores)
    if n == 0:
        return np.zeros(0, np.int64)
    if overlap not in ('polygon', 'mask'):
        raise ValueError(f"overlap must be 'polygon' or 'mask', got {overlap!r}")
    lib = hip.load()
    dev = torch.device('cuda', device)
    arrs = (np.ascontiguousarray(boxes, np.int32), np.ascontiguousarray(scores, np.float32), np.ascontiguousarray(areas, np.int32),
            np.ascontiguousarray(bits, np.uint32).view(np.int32), np.ascontiguousarray(off, np.int64))
    t = [torch.from_numpy(a).to(dev) for a in arrs]
    keep = torch.zeros(n, dtype=torch.uint8, device=dev)
    vp = lambda x: ctypes.c_void_p(x.data_ptr())
    b = arrs[0]
    with torch.cuda.device(dev):
        rc = lib.nuhtc_merge_overlap(device, vp(t[0]), vp(t[1]), vp(t[2]), vp(t[3]), vp(t[4]), n, int(arrs[3].size),
                                     hip.OVERLAP_POLYGON if overlap == 'polygon' else hip.OVERLAP_MASK, float(overlap_threshold),
                                     int(b[:, 0].min()), int(b[:, 1].min()), int(b[:, 2].max()), int(b[:, 3].max()),
                                     vp(keep), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc:
        raise RuntimeError(f'nuhtc_merge_overlap failed ({rc})')
    return np.nonzero(keep.cpu().numpy())[0].astype(np.int64)


def merge_overlap(rec, overlap_threshold=0.05, device=0, overlap='polygon'):
    """Cross-tile duplicate removal, strategy 'probability' of tools/nuclei_merge.py:62-174: detections sorted by
    score (descending), each still-alive one suppresses every later one it overlaps with IoU > threshold.
    overlap='polygon' (default) is the reference's measure: the IoU of the shapely polygons of the rings infer_wsi.py
    writes (first cv2 contour through the border-pixel centres, buffer(0) + largest part for self-touching rings), computed
    exactly on the GPU from the mask crops (nuhtc_merge_overlap, csrc/merge.hip); 'mask' is the IoU of the pixel sets.
    Returns kept indices (ascending)."""
    if len(rec['score']) == 0:
        return np.zeros(0, np.int64)
    boxes, areas, bits, off = pack_masks(rec['mask'])
    return merge_overlap_packed(boxes, rec['score'], areas, bits, off, overlap_threshold, device, overlap)


def pack_records(rec, keep=None, tile_base=0, rles=None):
    """Per-detection records of one rank -> the tensors that cross the node in the single gather (SURVEY §8e record layout):
    head  float64 (n, 9): box x0,y0,x1,y1 (slide px), score, label, ring length, tile index, RLE length
    verts int32   (sum ring lengths, 2): the closed rings, concatenated
    crops int64   (n, 6): mask-crop box x0,y0,x1,y1 (x1,y1 exclusive), set pixels, word offset into `bits`
    bits  int32   (words,): the bit-packed mask crops (the merge's input)
    blob  uint8   (bytes,): optional COCO RLE strings, concatenated"""
    import torch
    if isinstance(rec['mask'], PackedMasks) and rec['mask'].arrays is not None:          # records of the packed path: whole-slide array operations
        a = rec['mask'].arrays
        kp = np.arange(len(a['score']), dtype=np.int64) if keep is None else np.asarray(list(keep), np.int64)
        n = len(kp)
        head = np.zeros((n, 9), np.float64)
        vflat, vn = a['rings'].take(kp)
        head[:, :4], head[:, 4], head[:, 5], head[:, 6], head[:, 7] = a['box'][kp], a['score'][kp], a['label'][kp], vn, tile_base + a['tile'][kp]
        head[:, 8] = [len(r) for r in rles] if rles else 0
        verts = vflat.astype(np.int32).reshape(-1, 2)
        m = rec['mask'].subset(kp)
        crops = np.concatenate([m.boxes.astype(np.int64), m.areas[:, None].astype(np.int64), m.off[:, None]], 1) if n else np.zeros((0, 6), np.int64)
        blob = np.frombuffer(b''.join(rles), np.uint8).copy() if rles else np.zeros(0, np.uint8)
        return [torch.from_numpy(head), torch.from_numpy(verts), torch.from_numpy(crops), torch.from_numpy(m.bits.view(np.int32).copy()),
                torch.from_numpy(blob)]
    keep = list(range(len(rec['score']))) if keep is None else list(keep)
    n = len(keep)
    head = np.zeros((n, 9), np.float64)
    for k, i in enumerate(keep):
        head[k, :4] = rec['box'][i]
        head[k, 4], head[k, 5], head[k, 6] = rec['score'][i], rec['label'][i], len(rec['ring'][i])
        head[k, 7] = tile_base + rec['tile'][i]
        head[k, 8] = len(rles[k]) if rles else 0
    verts = np.concatenate([rec['ring'][i] for i in keep], 0).astype(np.int32) if n else np.zeros((0, 2), np.int32)
    mb, ma, mbits, moff = pack_masks([rec['mask'][i] for i in keep])
    crops = np.concatenate([mb.astype(np.int64), ma[:, None].astype(np.int64), moff[:, None]], 1) if n else np.zeros((0, 6), np.int64)
    if n == 0:
        mbits = np.zeros(0, np.uint32)
    blob = np.frombuffer(b''.join(rles), np.uint8).copy() if rles else np.zeros(0, np.uint8)
    return [torch.from_numpy(head), torch.from_numpy(verts), torch.from_numpy(crops), torch.from_numpy(mbits.view(np.int32).copy()),
            torch.from_numpy(blob)]


def merge_gathered(gathered, overlap_threshold=0.05, device=0, overlap='polygon'):
    """Rank 0 after the gather: `gathered` = per rank [head, verts, crops, bits, blob] (parallel.gather_blobs of pack_records)
    -> kept indices into the rank-major concatenation of the records (nuhtc_merge_overlap on this GPU)."""
    pk = [g[2].cpu().numpy() for g in gathered]
    wb = [g[3].cpu().numpy().view(np.uint32) for g in gathered]
    base = np.cumsum([0] + [len(w) for w in wb[:-1]])
    allp = np.concatenate(pk, 0)
    if len(allp) == 0:
        return np.zeros(0, np.int64)
    off_all = np.concatenate([p[:, 5] + b0 for p, b0 in zip(pk, base)])
    scores = np.concatenate([g[0].cpu().numpy()[:, 4] for g in gathered]).astype(np.float32)
    bits = np.concatenate(wb) if sum(len(w) for w in wb) else np.zeros(1, np.uint32)
    return merge_overlap_packed(allp[:, :4], scores, allp[:, 4], bits, off_all, overlap_threshold, device, overlap)
