"""Host-side driver of one libnuhtc_hip engine (one per GPU / process).

Plays the role of the reference's `HybridTaskCascade_Cus` module object (nuhtc/models/htc_cus.py): holds the
weights on the device and turns a batch of uint8 tiles into `(bbox_results, segm_results)` per tile, in the
exact format `inference_detector` returns (mmdet/apis/inference.py:90-153).  torch is used for
device buffers and the current stream only; every computation happens inside the HIP library.
"""
import ctypes
import os
import sys

import numpy as np
import torch

from . import hip


class HipError(RuntimeError):
    pass


class _DevView:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(int(s) for s in shape), typestr=typestr, data=(int(ptr), False),
                                             version=2, strides=None)


_TYPESTR = {0: '<f4', 1: '<i4', 2: '|u1', 3: '<u4'}
_TORCH = {0: torch.float32, 1: torch.int32, 2: torch.uint8, 3: torch.int32}


class Engine:
    EXPORT_BUFFERS = 3          # pinned host buffers of export_async: batches per pipeline slot (2) + 1, see export_async

    _bound = {}                 # thread id -> number of live engines that placed that thread (the last one to close restores its mask)
    _logged = False

    def __init__(self, state_dict, device=0, max_batch=16, tile=(256, 256), num_classes=5, bind_host=None, **cfg_overrides):
        if not torch.cuda.is_available():
            raise HipError('no HIP device visible: nuhtc_amd has no CPU path (the CPU oracle lives under oracle/ and is test-only)')
        self.lib = hip.load()
        self.device = torch.device('cuda', device if isinstance(device, int) else torch.device(device).index or 0)
        cfg = hip.default_config()
        cfg.num_classes = num_classes
        # `tile` is the image size (h, w), any size: the buffers the library sees are (h, w rounded up to a multiple of 32: bit-packed
        # mask rows) with the image in the top-left corner; like the reference's test pipeline the library resizes the image, pads the
        # network input to a multiple of 32 and clips boxes / pastes masks with the un-padded sizes (nuhtc_config.valid_h / valid_w)
        self.image_hw = (int(tile[0]), int(tile[1]))
        cfg.tile_h, cfg.tile_w = self.image_hw[0], -(-self.image_hw[1] // 32) * 32
        cfg.valid_h, cfg.valid_w = self.image_hw
        cfg.max_batch = int(max_batch)
        for k, v in cfg_overrides.items():
            if not hasattr(cfg, k):
                raise KeyError(f'unknown engine option {k}')
            if k == 'stage_stds':
                for i in range(3):
                    for j in range(4):
                        cfg.stage_stds[i][j] = float(v[i][j])
            elif k in ('mean', 'std'):
                for i in range(3):
                    getattr(cfg, k)[i] = float(v[i])
            else:
                setattr(cfg, k, v)
        self.cfg = cfg
        # The submitting thread belongs on the GPU's NUMA node (hip.bind_host_thread) -- before the first queue
        # exists.  Opt-in (bind_host=True, or NUHTC_HOST_AFFINITY=1 when the argument is None): narrowing the caller's CPU mask is
        # the caller's decision; close() gives the mask back when the last engine that placed this thread goes.
        self._placed_thread = None
        if bind_host is None:
            bind_host = os.environ.get('NUHTC_HOST_AFFINITY', '0') == '1'
        if bind_host:
            import threading
            try:
                before = os.sched_getaffinity(0)
                if hip.bind_host_thread(self.device.index):
                    tid = threading.get_ident()
                    Engine._bound[tid] = Engine._bound.get(tid, 0) + 1
                    self._placed_thread = tid
                    now = os.sched_getaffinity(0)
                    if not Engine._logged and now != before:
                        Engine._logged = True
                        print(f'nuhtc_amd: the submitting thread now runs on {len(now)} of its {len(before)} CPUs, the NUMA node of GPU {self.device.index} '
                              '(bind_host=True / NUHTC_HOST_AFFINITY=1; the mask is restored when the engine is closed)', file=sys.stderr)
                else:
                    import warnings
                    warnings.warn(f'nuhtc_amd: bind_host requested but the submitting thread was not placed: {hip.bind_reason}')
            except RuntimeError:          # placement is an optimisation: a device the runtime cannot name is reported by nuhtc_create below
                pass
        self.h = ctypes.c_void_p()
        rc = self.lib.nuhtc_create(ctypes.byref(cfg), self.device.index, ctypes.byref(self.h))
        if rc:
            raise HipError(f'nuhtc_create failed ({rc}): {self.lib.nuhtc_last_error(None).decode()}')
        from .weights import schema
        names = schema(num_classes)
        for name, t in state_dict.items():
            if name not in names:      # buffers / EMA / optimizer entries: not part of the path (the library rejects them)
                continue
            a = np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)
            if a.ndim == 0:
                a = a.reshape(1)
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.nuhtc_load_weight(self.h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim))
        self._check(self.lib.nuhtc_finalize(self.h))
        # a stream of the engine's own (nuhtc_stream): EnginePipeline runs the engine on it; any other stream works as well
        self.stream = torch.cuda.ExternalStream(self.lib.nuhtc_stream(self.h), device=self.device)
        B, K = cfg.max_batch, cfg.max_per_img
        # allocated and zero-filled ON the engine's stream (hipStreamNonBlocking: nothing would order memsets of the caller's stream
        # before the first batch there), so the blocks also live in the allocator pool of the stream they are used on
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            self.boxes = torch.zeros(B, K, 5, dtype=torch.float32, device=self.device)
            self.labels = torch.zeros(B, K, dtype=torch.int32, device=self.device)
            self.counts = torch.zeros(B, dtype=torch.int32, device=self.device)
            self.masks = torch.zeros(B, K, cfg.tile_h, cfg.tile_w // 32, dtype=torch.int32, device=self.device)
            self.areas = torch.zeros(B, K, dtype=torch.int32, device=self.device)
            self.keep = torch.zeros(B, K, dtype=torch.uint8, device=self.device)
        self.stream.synchronize()       # creation is synchronous anyway (nuhtc_finalize): the zero fills are complete whatever stream the caller uses
        self.dets = hip.Dets(self.boxes.data_ptr(), self.labels.data_ptr(), self.counts.data_ptr(), self.masks.data_ptr(),
                             self.areas.data_ptr(), self.keep.data_ptr())

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc:
            if not self.h:
                raise HipError('this engine was closed (Engine.close(), or evicted from the Detector cache of max_engines sizes): '
                               'ask the Detector for the engine again instead of keeping it')
            raise HipError(f'libnuhtc_hip error {rc}: {self.lib.nuhtc_last_error(self.h).decode()}')

    def close(self):
        if getattr(self, 'h', None) and self.h.value:
            self.lib.nuhtc_destroy(self.h)
            self.h = ctypes.c_void_p()
        tid = getattr(self, '_placed_thread', None)
        if tid is not None:
            self._placed_thread = None
            import threading
            Engine._bound[tid] = Engine._bound.get(tid, 1) - 1
            if Engine._bound[tid] <= 0:
                Engine._bound.pop(tid, None)
                if threading.get_ident() == tid:          # only the placed thread itself can take its mask back
                    hip.restore_host_thread()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, tiles):
        """(B,H,W,3) uint8 ndarray / tensor -> contiguous device tensor."""
        if isinstance(tiles, np.ndarray):
            tiles = torch.from_numpy(np.ascontiguousarray(tiles))
        if tiles.dtype != torch.uint8 or tiles.dim() != 4 or tiles.shape[-1] != 3:
            raise ValueError('tiles must be uint8 (B,H,W,3)')
        if tuple(tiles.shape[1:3]) != self.image_hw:
            raise ValueError(f'tile size {tuple(tiles.shape[1:3])} != engine tile size {self.image_hw}')
        tiles = tiles.to(self.device, non_blocking=True)
        if self.cfg.tile_w != self.image_hw[1]:          # row pitch of the library's buffers: width rounded up to 32 (content ignored)
            tiles = torch.nn.functional.pad(tiles, (0, 0, 0, self.cfg.tile_w - self.image_hw[1]))
        return tiles.contiguous()

    # ------------------------------------------------------------------ hot path
    def infer_async(self, tiles_dev, channel_mode=hip.CH_AS_IS):
        """Enqueue the whole path for a device-resident batch; outputs land in self.boxes/labels/counts/masks/keep."""
        B = tiles_dev.shape[0]
        self._last_tiles = tiles_dev          # (kept alive until the next call: buffer('img') is computed from them on request)
        self._check(self.lib.nuhtc_infer(self.h, ctypes.c_void_p(tiles_dev.data_ptr()), B, channel_mode, self._stream(),
                                         ctypes.byref(self.dets)))
        return B

    def infer_fixed_load_async(self, tiles_dev, rois_dev, n_dets, channel_mode=hip.CH_AS_IS):
        B, n_rois = tiles_dev.shape[0], rois_dev.shape[1]
        self._last_tiles = tiles_dev
        self._check(self.lib.nuhtc_infer_fixed_load(self.h, ctypes.c_void_p(tiles_dev.data_ptr()), B, channel_mode,
                                                    ctypes.c_void_p(rois_dev.data_ptr()), n_rois, n_dets, self._stream(),
                                                    ctypes.byref(self.dets)))
        return B

    def check(self):
        self._check(self.lib.nuhtc_check(self.h, self._stream()))

    def contours_async(self, B, cap=256, kept_only=True):
        """Enqueue the outer-contour trace (cv2.findContours(...)[0][0] of tools/infer_wsi.py:51-54) of the last infer's
        masks on the device; results in self.contour_xy (B,K,cap,2) int16 / self.contour_n (B,K) int32."""
        K = self.cfg.max_per_img
        if getattr(self, 'contour_xy', None) is None or self.contour_xy.shape[2] != cap:
            self.contour_xy = torch.zeros(self.cfg.max_batch, K, cap, 2, dtype=torch.int16, device=self.device)
            self.contour_n = torch.zeros(self.cfg.max_batch, K, dtype=torch.int32, device=self.device)
        dets = self.dets if kept_only else hip.Dets(self.boxes.data_ptr(), self.labels.data_ptr(), self.counts.data_ptr(),
                                                    self.masks.data_ptr(), self.areas.data_ptr(), None)
        self._check(self.lib.nuhtc_mask_contours(self.h, ctypes.byref(dets), B, cap, ctypes.c_void_p(self.contour_xy.data_ptr()),
                                                 ctypes.c_void_p(self.contour_n.data_ptr()), self._stream()))

    def contours(self, B, cap=256, kept_only=True):
        """-> per tile, dict slot -> (n,2) int64 open contour in tile pixels.  Contours that overflow the device
        capacities are traced by the host mirror (nuhtc_amd.contours) from the mask."""
        from . import contours as host
        self.contours_async(B, cap, kept_only)
        n = self.contour_n[:B].cpu().numpy()
        out = []
        for b in range(B):
            d = {}
            slots = np.nonzero(n[b])[0]
            if len(slots):
                xy = self.contour_xy[b, torch.from_numpy(slots).to(self.device)].cpu().numpy()
                for k, sl in enumerate(slots):
                    if n[b, sl] > 0:
                        d[int(sl)] = xy[k, :n[b, sl]].astype(np.int64)
                    else:
                        words = self.masks[b, sl].cpu().numpy().view(np.uint32)
                        bits = np.unpackbits(words.view(np.uint8).reshape(self.cfg.tile_h, self.cfg.tile_w // 8), axis=-1, bitorder='little')
                        d[int(sl)] = host.trace_outer_contour(bits.astype(bool))
            out.append(d)
        return out

    def export_async(self, B, cap=None, contour_cap=256, crop_words_per_det=128):
        """After infer_async: enqueue, on the current stream, everything the slide loop needs from the batch -- the outer
        contours (nuhtc_mask_contours), a gather of the kept detections, in (tile, slot) order (nuhtc_export_kept), and their
        masks cropped to their bounding rectangles into one word pool (nuhtc_export_crops) -- into fixed-capacity pinned host
        buffers.  No host synchronisation: every device -> host copy of a result would otherwise block the submitting thread
        behind the other batches in flight.  Read with export_read() once the stream (or an event recorded after this call) has
        completed.  The full 8 KB masks stay on the device (export_full_mask fetches one when a crop did not fit the pool)."""
        K, W = self.cfg.max_per_img, self.cfg.tile_h * (self.cfg.tile_w // 32)
        # default capacity: 96 kept detections per tile on average at the x2 resize of a 40x slide, scaled with the nuclei per tile at
        # larger factors (20x slides: x4 -> four times the nuclei per 256-px tile); a batch over it falls back for that batch only
        per_tile = 96 * max(1, int(round((float(self.cfg.scale_factor) / 2.0) ** 2)))
        cap = int(cap or min(self.cfg.max_batch * K, per_tile * self.cfg.max_batch))
        pool = cap * int(crop_words_per_det)
        ex = getattr(self, '_ex', None)
        if ex is None or ex['cap'] != cap or ex['ccap'] != contour_cap or ex['pool'] != pool:
            dev = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=self.device)
            names = dict(nk=((2,), torch.int32), idx=((cap,), torch.int64), boxes=((cap, 5), torch.float32), labels=((cap,), torch.int32),
                         cn=((cap,), torch.int32), crop_box=((cap, 4), torch.int32), crop_area=((cap,), torch.int32),
                         crop_off=((cap + 1,), torch.int32), crop_words=((pool,), torch.int32), xy=((cap, contour_cap, 2), torch.int16))
            # every field is a view into ONE device buffer and ONE pinned host buffer: a batch's results leave the device in a single
            # copy (each asynchronous copy on a compute stream costs a hand-over between the copy engine and the kernels)
            offs, total = {}, 0
            for k, (sh, dt) in names.items():
                offs[k] = total
                total += (int(np.prod(sh)) * torch.empty(0, dtype=dt).element_size() + 255) // 256 * 256
            blob_dev = torch.zeros(total, dtype=torch.uint8, device=self.device)
            # EXPORT_BUFFERS = 3 pinned host buffers, used in turn.  A pipeline slot holds up to two exported batches (A running or
            # done, B queued behind it); when A has been collected and the slot is resubmitted (C) BEFORE A's views are unpacked, C's
            # copy must not land in A's buffer nor in B's: three buffers.  The views of a collected batch stay intact until the second
            # export_async after the one that filled them (i.e. until the slot's next-but-one batch is enqueued)
            blob_hosts = [torch.zeros(total, dtype=torch.uint8).pin_memory() for _ in range(self.EXPORT_BUFFERS)]
            view = lambda blob, k: blob[offs[k]:offs[k] + int(np.prod(names[k][0])) * torch.empty(0, dtype=names[k][1]).element_size()].view(names[k][1]).view(*names[k][0])
            ex = self._ex = dict(cap=cap, ccap=contour_cap, pool=pool, blob_dev=blob_dev, blob_hosts=blob_hosts, turn=0,
                                 hosts=[{k: view(b, k) for k in names} for b in blob_hosts], dev={k: view(blob_dev, k) for k in names})
            ex['dev']['words'] = dev(cap, W, dtype=torch.int32)          # full masks of the kept detections: device only
        self.contours_async(B, contour_cap)
        d = ex['dev']
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        # compaction of the kept detections on the device (nuhtc_export_kept), then one asynchronous copy per field
        self._check(self.lib.nuhtc_export_kept(self.h, ctypes.byref(self.dets), B, vp(self.contour_n), vp(self.contour_xy), contour_cap, cap,
                                               vp(d['nk']), vp(d['idx']), vp(d['boxes']), vp(d['labels']), vp(d['cn']), vp(d['xy']), vp(d['words']),
                                               self._stream()))
        self._check(self.lib.nuhtc_export_crops(self.h, vp(d['words']), vp(d['nk']), cap, vp(d['crop_box']), vp(d['crop_area']), vp(d['crop_off']),
                                                vp(d['crop_words']), pool, self._stream()))
        ex['turn'] = (ex['turn'] + 1) % self.EXPORT_BUFFERS
        ex['host'] = ex['hosts'][ex['turn']]
        ex['blob_hosts'][ex['turn']].copy_(ex['blob_dev'], non_blocking=True)
        ex['B'] = B
        self._read_turn = None          # (EnginePipeline.collect points export_read() at the buffer of the batch it returns)
        return ex['turn']

    def export_read(self, turn=None):
        """-> dict of numpy views (n kept detections: tile index in the batch, slot, box+score, label, contour length
        (<= 0: traced by the host mirror), contour vertices, bit-packed mask words) of the pinned buffers export_async
        filled, or None when the batch held more kept detections than the buffers (use the synchronous path then).
        The views stay valid through the next TWO export_async calls of this engine (EXPORT_BUFFERS = 3 host buffers used in turn:
        a slot of an EnginePipeline may be resubmitted before the batch it just delivered is unpacked), not the one after those;
        `turn` selects the buffer of an earlier export_async (its return value) when the next batch has been enqueued already.
        Raises on the capacity flag of that inference (what check() reports)."""
        if turn is None:
            turn = getattr(self, '_read_turn', None)
        ex = self._ex['hosts'][self._ex['turn'] if turn is None else turn]
        if int(ex['nk'][1]):
            raise HipError('connected-component proposals exceeded max_cc_proposals on at least one tile (NUHTC_E_CAPACITY)')
        n = int(ex['nk'][0])
        if n > self._ex['cap']:
            return None
        K = self.cfg.max_per_img
        idx = ex['idx'][:n].numpy()
        off = ex['crop_off'].numpy()
        return dict(n=n, tile=idx // K, slot=idx % K, boxes=ex['boxes'][:n].numpy(), labels=ex['labels'][:n].numpy(), cn=ex['cn'][:n].numpy(),
                    xy=ex['xy'][:n].numpy(), crop_box=ex['crop_box'][:n].numpy(), crop_area=ex['crop_area'][:n].numpy(), crop_off=off[:n],
                    crop_words=ex['crop_words'].numpy().view(np.uint32), crop_total=int(off[self._ex['cap']]), pool=self._ex['pool'])

    def export_full_mask(self, k):
        """(tile_h, tile_w) bool mask of exported detection k of the last export_async (synchronous device read: the rare crop that
        did not fit the pool, or a contour the device could not trace)."""
        words = self._ex['dev']['words'][k].cpu().numpy().view(np.uint32)
        return np.unpackbits(words.view(np.uint8).reshape(self.cfg.tile_h, self.cfg.tile_w // 8), axis=-1, bitorder='little').astype(bool)

    def results(self, B, with_masks=True):
        """Device outputs of the last infer -> list of (bbox_results, segm_results) exactly like the reference
        (`bbox2result` mmdet/core/bbox/transforms.py:100-117; `get_seg_masks` list-of-bool-arrays per class)."""
        self.check()
        counts = self.counts[:B].cpu().numpy()
        boxes = self.boxes[:B].cpu().numpy()
        labels = self.labels[:B].cpu().numpy()
        nc = self.cfg.num_classes
        H, W = self.cfg.tile_h, self.cfg.tile_w
        out = []
        for b in range(B):
            n = int(counts[b])
            d, l = boxes[b, :n], labels[b, :n]
            bbox_res = [d[l == c] for c in range(nc)]
            if with_masks and n:
                words = self.masks[b, :n].cpu().numpy().view(np.uint32)
                bits = np.unpackbits(words.view(np.uint8).reshape(n, H, W // 8), axis=-1, bitorder='little').astype(bool)[..., :self.image_hw[1]]
                segm_res = [[bits[j] for j in range(n) if l[j] == c] for c in range(nc)]
            else:
                segm_res = [[] for _ in range(nc)]
            out.append((bbox_res, segm_res))
        return out

    def __call__(self, tiles, channel_mode=hip.CH_AS_IS):
        t = self.to_device(tiles)
        out = []
        for i in range(0, t.shape[0], self.cfg.max_batch):
            chunk = t[i:i + self.cfg.max_batch]
            B = self.infer_async(chunk, channel_mode)
            out.extend(self.results(B))
        return out

    # ------------------------------------------------------------------ parity-test access
    def enable_token_dump(self):
        p = ctypes.c_void_p()
        self._check(self.lib.nuhtc_get_buffer(self.h, b'__enable_token_dump', ctypes.byref(p), None, None, None))

    def buffer(self, name):
        """Copy of an intermediate tensor of the last infer call (see nuhtc_get_buffer)."""
        p = ctypes.c_void_p()
        shape = (ctypes.c_int64 * 6)()
        nd, dt = ctypes.c_int(), ctypes.c_int()
        self._check(self.lib.nuhtc_get_buffer(self.h, name.encode(), ctypes.byref(p), shape, ctypes.byref(nd), ctypes.byref(dt)))
        torch.cuda.current_stream(self.device).synchronize()
        shp = [shape[i] for i in range(nd.value)]
        view = torch.as_tensor(_DevView(p.value, shp, _TYPESTR[dt.value]), device=self.device)
        return view.clone()

    def op_gemm(self, A, W, bias=None, act=0, pipe='fp32'):
        """C = act(A @ W.T + bias) by the engine's GEMM kernel; pipe 'fp32' (v_mfma_f32_32x32x2_f32) or 'split' (exact 3-way bf16
        split of both operands on v_mfma_f32_32x32x16_bf16)."""
        M, K = A.shape
        N = W.shape[0]
        C = torch.empty(M, N, dtype=torch.float32, device=self.device)
        bp = bias.data_ptr() if bias is not None else None
        if pipe == 'fp32':
            self._check(self.lib.nuhtc_op_gemm(self.h, A.data_ptr(), W.data_ptr(), bp, C.data_ptr(), M, N, K, act, self._stream()))
        else:
            wh = np.ascontiguousarray(W.detach().cpu().numpy(), dtype=np.float32)
            self._check(self.lib.nuhtc_op_gemm_split(self.h, A.data_ptr(), W.data_ptr(), wh.ctypes.data_as(ctypes.c_void_p), bp, C.data_ptr(),
                                                     M, N, K, act, self._stream()))
        return C

    def op_ln_gemm(self, x, w, bias, ln_g, ln_b, rows=None, act=0):
        """act(LayerNorm(x[rows]) @ w.T + bias) with the norm in the product's A path (csrc/gemm.hip A_LN, statistics by ln_stats_kernel);
        x (T, K) on the device, rows: optional device int32 (M,) row indices, the rest anywhere."""
        T, K = x.shape
        M = int(rows.shape[0]) if rows is not None else T
        N = w.shape[0]
        out = torch.empty(M, N, dtype=torch.float32, device=self.device)
        h = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        wh, gh, bh = h(w), h(ln_g), h(ln_b)
        bias_h = h(bias) if bias is not None else None
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
        self._check(self.lib.nuhtc_op_ln_gemm(self.h, x.data_ptr(), T, rows.data_ptr() if rows is not None else None, vp(wh), vp(bias_h), vp(gh), vp(bh),
                                              out.data_ptr(), M, N, K, act, self._stream()))
        return out

    def op_gemm_ln_gemm(self, a, wp, bp, w, bias, ln_g, ln_b, res=None, row_map=None, act=0):
        """y[row_map] = a @ wp.T + bp (+ res[row_map]); c = act(LayerNorm(y) @ w.T + bias) the way the engine chains them: the producer's
        epilogue leaves the row statistics per 96 columns, the consumer merges them (csrc/gemm.hip stats_out / A_LN).  -> (y, c)."""
        M, Kp = a.shape
        K, N = wp.shape[0], w.shape[0]
        y = torch.zeros(M, K, dtype=torch.float32, device=self.device)
        c = torch.empty(M, N, dtype=torch.float32, device=self.device)
        h = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32) if t is not None else None
        wph, bph, wh, bh, gh, lbh = h(wp), h(bp), h(w), h(bias), h(ln_g), h(ln_b)
        vp = lambda x: x.ctypes.data_as(ctypes.c_void_p) if x is not None else None
        self._check(self.lib.nuhtc_op_gemm_ln_gemm(self.h, a.data_ptr(), vp(wph), vp(bph), res.data_ptr() if res is not None else None,
                                                   row_map.data_ptr() if row_map is not None else None, vp(wh), vp(bh), vp(gh), vp(lbh),
                                                   y.data_ptr(), c.data_ptr(), M, Kp, K, N, act, self._stream()))
        return y, c

    def op_merge_ln_gemm(self, x, B, H, W, w, ln_g, ln_b):
        """PatchMerging in one launch (csrc/gemm.hip A_LN over two segments per row): x (B*H*W, C) tokens on the device; w (2C, 4C), ln_g / ln_b (4C)
        in the reference's nn.Unfold column order (transformer.py:363-385).  -> (B*H/2*W/2, 2C)."""
        C = x.shape[1]
        y = torch.empty(B * (H // 2) * (W // 2), 2 * C, dtype=torch.float32, device=self.device)
        h = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        wh, gh, bh = h(w), h(ln_g), h(ln_b)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self.lib.nuhtc_op_merge_ln_gemm(self.h, x.data_ptr(), B, H, W, C, vp(wh), vp(gh), vp(bh), y.data_ptr(), self._stream()))
        return y

    def op_swin_mlp(self, x, ln_g, ln_b, w1, b1, w2, b2):
        """x + W2 gelu(W1 LN(x) + b1) + b2 by the fused FFN kernel (csrc/mlp.hip); x (T, C) and the vectors on the device, w1 / w2 anywhere."""
        T, C = x.shape
        out = torch.empty_like(x)
        w1h = np.ascontiguousarray(w1.detach().cpu().numpy(), dtype=np.float32)
        w2h = np.ascontiguousarray(w2.detach().cpu().numpy(), dtype=np.float32)
        self._check(self.lib.nuhtc_op_swin_mlp(self.h, x.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), w1h.ctypes.data_as(ctypes.c_void_p), b1.data_ptr(),
                                               w2h.ctypes.data_as(ctypes.c_void_p), b2.data_ptr(), out.data_ptr(), T, C, self._stream()))
        return out

    def op_swin_proj_mlp(self, x, att, wp, bp, ln_g, ln_b, w1, b1, w2, b2):
        """x' = x + Wp att + bp; x' + W2 gelu(W1 LN(x') + b1) + b2: the fused second half of a Swin block (csrc/mlp.hip with the attention
        projection in front); x, att (T, C) and the vectors on the device, the weight matrices anywhere."""
        T, C = x.shape
        out = torch.empty_like(x)
        h = lambda w: np.ascontiguousarray(w.detach().cpu().numpy(), dtype=np.float32)
        wph, w1h, w2h = h(wp), h(w1), h(w2)
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self._check(self.lib.nuhtc_op_swin_proj_mlp(self.h, x.data_ptr(), att.data_ptr(), cp(wph), bp.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), cp(w1h),
                                                    b1.data_ptr(), cp(w2h), b2.data_ptr(), out.data_ptr(), T, C, self._stream()))
        return out

    def op_roi_align(self, feat_nhwc, rois, P, scale, sr):
        N, H, W, C = feat_nhwc.shape
        R = rois.shape[0]
        out = torch.empty(R, P, P, C, dtype=torch.float32, device=self.device)
        self._check(self.lib.nuhtc_op_roi_align(self.h, feat_nhwc.data_ptr(), N, H, W, rois.data_ptr(), R, P, float(scale), int(sr),
                                                out.data_ptr(), self._stream()))
        return out

    def op_nms(self, boxes, scores, thr):
        n = boxes.shape[0]
        keep = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._check(self.lib.nuhtc_op_nms(self.h, boxes.data_ptr(), scores.data_ptr(), n, float(thr), keep.data_ptr(), cnt.data_ptr(),
                                          self._stream()))
        return keep[:int(cnt.item())].long()
