"""dev (-DNUHTC_DEV build): which allocation carries the fast / slow state of the 96-column GEMMs?  Creates E engines one after the other; for each
measures gemm_kernel<3> per step, then gives the six Swin activation buffers new memory ONE AT A TIME (nuhtc_dev_realloc) and measures after each.
usage: r04_state_buffers.py [E] [passes]"""
import ctypes, gc, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
E = int(sys.argv[1]) if len(sys.argv) > 1 else 3
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NAMES = ['tokA', 'tokB', 'xw', 'qkv', 'att', 'hid']
sd = weights.bench_state_dict()
tiles_h = synth.nuclei_tiles(16, 256)
def measure(eng, tiles, warm=12):
    for _ in range(warm): eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize()
    hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    shapes = {}
    for k, v in p.items():
        if k.startswith('gemm_kernel<3>'): shapes[k.split('>')[1]] = v['ms'] / 5
    return sum(shapes.values()), shapes
lib = hip.load()
lib.nuhtc_dev_realloc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
keep = []
for r in range(E):
    eng = Engine(sd, device=0, max_batch=16, tile=(256, 256))
    torch.cuda.set_stream(eng.stream)
    tiles = eng.to_device(tiles_h)
    g, sh0 = measure(eng, tiles, 30)
    print(f'engine {r}: gemm3 {g:.3f} ms', flush=True)
    for ps in range(PASSES):
        for w, n in enumerate(NAMES):
            a = ctypes.c_ulonglong()
            rc = lib.nuhtc_dev_realloc(eng.h, w, ctypes.byref(a))
            assert rc == 0, rc
            g2, sh = measure(eng, tiles)
            moved = sorted(((sh[k] - sh0[k]) * 1e3, k) for k in sh if abs(sh[k] - sh0[k]) > 0.004)
            print(f'   new {n:5s} @ {a.value:#x}: gemm3 {g2:.3f} ms ({(g2 - g) * 1e3:+.0f} us)  shapes that moved (us): ' + ' '.join(f'{k}:{d:+.0f}' for d, k in moved), flush=True)
            g, sh0 = g2, sh
    keep.append(eng)      # stays allocated: the next engine gets other memory
