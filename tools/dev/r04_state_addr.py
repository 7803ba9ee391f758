"""dev: is the fast / slow state of the dense launches tied to where the engine's buffers landed?  One process: creates an engine, runs steps,
prints the 96-column GEMM time per step beside the device addresses of a few engine buffers (modulo 2 MiB, 1 GiB) -- run it N times and look."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
mode = os.environ.get('STATE_MODE', 'alone')       # alone | prealloc (24 GB of torch memory first) | engines (four other engines first) | after (four engines created after it)
keep = []
if mode == 'prealloc':
    keep = [torch.empty(6 << 30, dtype=torch.uint8, device='cuda') for _ in range(4)]
if mode == 'engines':
    keep = [Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256)) for _ in range(4)]
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
if mode == 'after':
    keep = [Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256)) for _ in range(4)]
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
for _ in range(40): eng.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize()
# what else bench.py does before it reads the per-kernel figures: STATE_DO = comma list of export, probe, check, long, fixed, alloc
do = [x for x in os.environ.get('STATE_DO', '').split(',') if x]
import time
if 'export' in do:
    eng.export_async(16); torch.cuda.current_stream().synchronize(); eng.export_read()
if 'check' in do:
    eng.check(); _ = eng.counts[:16].cpu(); _ = eng.buffer('roi_counts')
if 'probe' in do:
    for st in [torch.cuda.Stream(device=tiles.device) for _ in range(3)]:
        pr = hip.ClockProbe(0, stream=st); torch.cuda.synchronize(); pr.start(100.0)
        for _ in range(10): eng.infer_async(tiles, hip.CH_SWAP)
        pr.ghz(); torch.cuda.synchronize()
if 'long' in do:
    t0 = time.time()
    while time.time() - t0 < 6.0:
        for _ in range(10): eng.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize()
if 'fixed' in do:
    rois = torch.from_numpy(synth.fixed_load_rois(16, size=(40.0, 100.0))).to(tiles.device)
    for _ in range(10): eng.infer_fixed_load_async(tiles, rois, 64, hip.CH_SWAP)
    torch.cuda.synchronize()
if 'alloc' in do:
    junk = [torch.zeros(64 << 20, dtype=torch.uint8).pin_memory() for _ in range(3)]
    junk2 = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); del junk2
for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize()
hip.profile_enable(True)
for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
p = hip.profile_read(); hip.profile_enable(False)
g3 = sum(v['ms'] for k, v in p.items() if k.startswith('gemm_kernel<3>')) / 5
ptrs = {}
for name in ('tokens', 'img', 'c0', 'c2', 'x0', 'rois', 'bbox_feats'):
    q = ctypes.c_void_p()
    eng._check(eng.lib.nuhtc_get_buffer(eng.h, name.encode(), ctypes.byref(q), None, None, None))
    ptrs[name] = q.value
print(mode, ','.join(do) or '-', 'gemm3 %.3f ms' % g3, ' '.join('%s=%#x(2M:%#x)' % (k, v, v & ((2 << 20) - 1)) for k, v in ptrs.items()), 'tiles=%#x' % tiles.data_ptr())
