"""dev: what re-rolls the fast / slow state of the 96-column GEMMs inside a process?  One engine; between measurements of gemm_kernel<3> per step:
A nothing | B hipMalloc of an unrelated 64 MiB buffer (kept) | C hipMalloc + hipFree | D hipMemset of an unrelated buffer | E 0.3 s of sleep |
F hipMalloc of 4 KiB (kept) | G hipMalloc of 1 GiB (kept).   usage: r04_state_trigger.py [iterations per phase]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rt = ctypes.CDLL('libamdhip64.so')
rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
rt.hipFree.argtypes = [ctypes.c_void_p]
rt.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
def measure(warm=12):
    for _ in range(warm): eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize()
    hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    return sum(v['ms'] for k, v in p.items() if k.startswith('gemm_kernel<3>')) / 5
def malloc(n):
    p = ctypes.c_void_p()
    assert rt.hipMalloc(ctypes.byref(p), n) == 0
    return p
scratch = malloc(64 << 20)
kept = []
def phase(name, fn):
    out = []
    for _ in range(IT):
        torch.cuda.synchronize()
        fn()
        out.append(measure())
    print(f'{name}: ' + ' '.join(f'{g:.3f}' for g in out), flush=True)
measure(30)
phase('A nothing           ', lambda: None)
phase('B malloc 64 MiB kept', lambda: kept.append(malloc(64 << 20)))
phase('A nothing           ', lambda: None)
phase('C malloc + free     ', lambda: rt.hipFree(malloc(64 << 20)))
phase('D memset unrelated  ', lambda: (rt.hipMemset(scratch, 0, 64 << 20), torch.cuda.synchronize()))
phase('E sleep 0.3 s       ', lambda: time.sleep(0.3))
phase('F malloc 4 KiB kept ', lambda: kept.append(malloc(4096)))
phase('G malloc 1 GiB kept ', lambda: kept.append(malloc(1 << 30)))
phase('A nothing           ', lambda: None)
