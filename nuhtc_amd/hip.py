"""ctypes binding of libnuhtc_hip.so — the C ABI declared in include/nuhtc_hip.h.

There is no fallback: if the shared library is missing or fails to load, importing callers get an
ImportError/OSError that says how to build it (`python -m nuhtc_amd.build`).
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libnuhtc_hip.so')

OK, E_INVALID, E_HIP, E_STATE, E_CAPACITY, E_NOTFOUND = 0, -1, -2, -3, -4, -5
CH_AS_IS, CH_SWAP = 0, 1
OVERLAP_MASK, OVERLAP_POLYGON = 0, 1
PIPE_BF16_SPLIT, PIPE_FP32 = 0, 1
SCHED_LATENCY, SCHED_THROUGHPUT = 0, 1


class Config(ctypes.Structure):
    _fields_ = [
        ('abi_version', ctypes.c_int32), ('num_classes', ctypes.c_int32),
        ('tile_h', ctypes.c_int32), ('tile_w', ctypes.c_int32), ('valid_h', ctypes.c_int32), ('valid_w', ctypes.c_int32), ('max_batch', ctypes.c_int32),
        ('scale_factor', ctypes.c_float), ('mean', ctypes.c_float * 3), ('std', ctypes.c_float * 3),
        ('rpn_nms_pre', ctypes.c_int32), ('rpn_max_per_img', ctypes.c_int32),
        ('rpn_nms_iou', ctypes.c_float), ('rpn_min_bbox_size', ctypes.c_float),
        ('score_thr', ctypes.c_float), ('nms_iou', ctypes.c_float), ('max_per_img', ctypes.c_int32),
        ('mask_thr_binary', ctypes.c_float), ('att_thres', ctypes.c_float),
        ('watershed_proposal', ctypes.c_int32), ('max_cc_proposals', ctypes.c_int32),
        ('stage_stds', (ctypes.c_float * 4) * 3),
        ('margin', ctypes.c_int32), ('min_area', ctypes.c_int32), ('mask_nms_thr', ctypes.c_float),
        ('matrix_pipe', ctypes.c_int32), ('schedule', ctypes.c_int32), ('att_pool_fp16', ctypes.c_int32),
    ]


class Dets(ctypes.Structure):
    _fields_ = [('boxes', ctypes.c_void_p), ('labels', ctypes.c_void_p), ('counts', ctypes.c_void_p),
                ('masks', ctypes.c_void_p), ('areas', ctypes.c_void_p), ('keep', ctypes.c_void_p)]


EXPORTS = ['nuhtc_default_config', 'nuhtc_create', 'nuhtc_destroy', 'nuhtc_last_error', 'nuhtc_load_weight',
           'nuhtc_finalize', 'nuhtc_infer', 'nuhtc_infer_fixed_load', 'nuhtc_check', 'nuhtc_get_buffer',
           'nuhtc_op_gemm', 'nuhtc_op_gemm_split', 'nuhtc_op_roi_align', 'nuhtc_op_nms', 'nuhtc_profile_enable', 'nuhtc_profile_read', 'nuhtc_dev_knob', 'nuhtc_export_crops',
           'nuhtc_mask_contours', 'nuhtc_merge_overlap', 'nuhtc_export_kept', 'nuhtc_clock_probe', 'nuhtc_op_swin_mlp', 'nuhtc_stream', 'nuhtc_op_swin_proj_mlp', 'nuhtc_bind_host_thread',
           'nuhtc_bind_host_thread_pci', 'nuhtc_bind_host_thread_at', 'nuhtc_restore_host_thread', 'nuhtc_op_ln_gemm', 'nuhtc_op_gemm_ln_gemm', 'nuhtc_op_merge_ln_gemm', 'nuhtc_write_ring_features',
           'nuhtc_write_point_features', 'nuhtc_join_features', 'nuhtc_fill_rings']

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback exists). '
                          'Build it with `python -m nuhtc_amd.build` (hipcc, --offload-arch=gfx950).')
    # torch first: it brings its own copy of the HIP runtime, and a process must hold ONE -- loaded after torch, this library's libamdhip64
    # dependency resolves to the copy torch loaded; loaded before it (e.g. __graft_entry__.build() followed by smoke() in one process), the
    # process ends up with two runtimes and the second one finds no device (nuhtc_create: "no such HIP device")
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.nuhtc_default_config.argtypes = [ctypes.POINTER(Config)]
    lib.nuhtc_default_config.restype = None
    lib.nuhtc_create.argtypes = [ctypes.POINTER(Config), ci, ctypes.POINTER(vp)]
    lib.nuhtc_destroy.argtypes = [vp]
    lib.nuhtc_destroy.restype = None
    lib.nuhtc_last_error.argtypes = [vp]
    lib.nuhtc_last_error.restype = ctypes.c_char_p
    lib.nuhtc_load_weight.argtypes = [vp, ctypes.c_char_p, vp, ctypes.POINTER(ctypes.c_int64), ci]
    lib.nuhtc_finalize.argtypes = [vp]
    lib.nuhtc_infer.argtypes = [vp, vp, ci, ci, vp, ctypes.POINTER(Dets)]
    lib.nuhtc_infer_fixed_load.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp, ctypes.POINTER(Dets)]
    lib.nuhtc_check.argtypes = [vp, vp]
    lib.nuhtc_get_buffer.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64),
                                     ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.nuhtc_op_gemm.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.nuhtc_op_gemm_split.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.nuhtc_op_ln_gemm.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.nuhtc_op_gemm_ln_gemm.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.nuhtc_op_merge_ln_gemm.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.nuhtc_op_swin_mlp.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.nuhtc_op_swin_proj_mlp.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.nuhtc_op_roi_align.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, cf, ci, vp, vp]
    lib.nuhtc_op_nms.argtypes = [vp, vp, vp, ci, cf, vp, vp, vp]
    lib.nuhtc_mask_contours.argtypes = [vp, ctypes.POINTER(Dets), ci, ci, vp, vp, vp]
    lib.nuhtc_merge_overlap.argtypes = [ci, vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ci, ctypes.c_double, ci, ci, ci, ci, vp, vp]
    lib.nuhtc_export_kept.argtypes = [vp, ctypes.POINTER(Dets), ci, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nuhtc_export_crops.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, ci, vp]
    lib.nuhtc_profile_enable.argtypes = [ci]
    lib.nuhtc_clock_probe.argtypes = [ci, ctypes.c_uint64, vp, vp]
    lib.nuhtc_dev_knob.argtypes = [ctypes.c_char_p, ci]
    lib.nuhtc_bind_host_thread.argtypes = [ci]
    lib.nuhtc_bind_host_thread_pci.argtypes = [ctypes.c_char_p]
    lib.nuhtc_bind_host_thread_at.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.nuhtc_restore_host_thread.argtypes = []
    lib.nuhtc_stream.argtypes = [vp]
    lib.nuhtc_profile_read.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    pcp = ctypes.POINTER(ctypes.c_char_p)
    lib.nuhtc_write_ring_features.argtypes = [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_char_p, pcp, pcp, ctypes.c_int32, vp, ctypes.c_int64, vp, ctypes.c_int32]
    lib.nuhtc_write_point_features.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_char_p, pcp, pcp, ctypes.c_int32, vp, ctypes.c_int64]
    lib.nuhtc_join_features.argtypes = [vp, vp, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int32]
    lib.nuhtc_fill_rings.argtypes = [vp, vp, ctypes.c_int64, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int32]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is ctypes.c_int or name not in ('nuhtc_default_config', 'nuhtc_destroy', 'nuhtc_last_error', 'nuhtc_stream'):
            fn.restype = ci
    lib.nuhtc_last_error.restype = ctypes.c_char_p
    lib.nuhtc_default_config.restype = None
    lib.nuhtc_destroy.restype = None
    lib.nuhtc_stream.restype = ctypes.c_void_p
    for fn in (lib.nuhtc_write_ring_features, lib.nuhtc_write_point_features, lib.nuhtc_join_features, lib.nuhtc_fill_rings):
        fn.restype = ctypes.c_int64
    _lib = lib
    return lib


def default_config():
    cfg = Config()
    load().nuhtc_default_config(ctypes.byref(cfg))
    return cfg


BIND_REASON = {0: 'bound', -1: 'not a PCI address', -2: 'no such HIP device', -3: "the thread's own CPU mask excludes the GPU's NUMA node",
               -5: 'the host exposes no NUMA node for the device'}
bind_reason = 'never called'          # why the last bind_host_thread of this process returned what it did (BIND_REASON)


def bind_host_thread(device=0, pci_bdf=None, sysfs_root=None):
    """Restrict the calling thread to the CPUs of the NUMA node the GPU is attached to (nuhtc_bind_host_thread: the thread that submits
    an engine's work should run there -- from the other socket every dispatch packet costs the command processor 1.4-2.9 us more),
    intersected with the mask the thread had before its first placement (`restore_host_thread` gives that mask back).
    Returns True when the thread now runs inside that node, False when nothing was changed; `hip.bind_reason` then says why (no NUMA
    node for the device / the caller's own mask excludes it / not a PCI address).  A device the runtime cannot name raises.
    `sysfs_root`: the test entry point (nuhtc_bind_host_thread_at)."""
    global bind_reason
    lib = load()
    if sysfs_root is not None:
        rc = lib.nuhtc_bind_host_thread_at(str(sysfs_root).encode(), pci_bdf.encode())
    else:
        rc = lib.nuhtc_bind_host_thread_pci(pci_bdf.encode()) if pci_bdf else lib.nuhtc_bind_host_thread(int(device))
    bind_reason = BIND_REASON.get(rc, f'error {rc}')
    if rc == -2:
        raise RuntimeError(f'nuhtc_bind_host_thread: no such device ({device})')
    return rc == 0


def restore_host_thread():
    """The calling thread gets back the CPU mask it had before its first bind_host_thread (no-op when it was never placed)."""
    return load().nuhtc_restore_host_thread() == 0


def profile_enable(on=True):
    load().nuhtc_profile_enable(1 if on else 0)


def dev_knob(name, value):
    """Development: set a switch of the launch heuristics (NUHTC_<NAME>) at run time."""
    rc = load().nuhtc_dev_knob(name.encode(), int(value))
    if rc:
        raise RuntimeError('nuhtc_dev_knob: not a development build (NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force)')


def profile_read():
    """{tag: dict(launches, ms, flops, bytes)} of the kernels launched since profile_enable / the last read."""
    buf = ctypes.create_string_buffer(1 << 16)
    rc = load().nuhtc_profile_read(buf, len(buf))
    if rc:
        raise RuntimeError(f'nuhtc_profile_read failed ({rc})')
    out = {}
    for line in buf.value.decode().splitlines():
        tag, n, ms, fl, by = line.rsplit(' ', 4)
        out[tag] = dict(launches=int(n), ms=float(ms), flops=float(fl), bytes=float(by))
    return out


class ClockProbe:
    """Shader clock held while other kernels run: start() enqueues the one-wave probe for `ms` milliseconds, ghz() waits for it and
    returns cycles / reference ticks x 0.1 GHz.  The probe runs on a NON-BLOCKING stream of its own (never the legacy null stream:
    a kernel spinning there serialises every blocking stream of the process, torch's default stream included, for the whole probe).
    Create the probe AFTER the engines: streams are dealt round the runtime's hardware queues in creation order, and a probe that
    shares a queue with an engine runs alone, ahead of the steps it is meant to run beside (it then reports the idle clock: with
    four one-stream engines on the default four queues that cannot be avoided, and ghz() is then the clock between their batches)."""

    def __init__(self, device=0, stream=None):
        import torch
        self.device = device
        if stream is None:          # torch's pool streams are created with hipStreamNonBlocking
            stream = torch.cuda.Stream(device=torch.device('cuda', device))
        self.stream = stream
        with torch.cuda.stream(self.stream):
            self.out = torch.zeros(2, dtype=torch.int64, device=torch.device('cuda', device))
        self.stream.synchronize()

    def start(self, ms):
        rc = load().nuhtc_clock_probe(self.device, int(ms * 1e5), ctypes.c_void_p(self.out.data_ptr()), ctypes.c_void_p(self.stream.cuda_stream))
        if rc:
            raise RuntimeError(f'nuhtc_clock_probe failed ({rc})')

    def ghz(self):
        self.stream.synchronize()
        with __import__('torch').cuda.stream(self.stream):
            c, r = (int(v) for v in self.out.cpu())
        return 0.1 * c / r if r else None
