"""Dev: the Swin linears of a process run at ~4.4 or ~4.8 ms per step.  Per process: addresses of some buffers, duration of the
dominant kernel for two engines created one after the other (is the state per process or per allocation?), probe clock."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sd = weights.bench_state_dict()
tiles_np = synth.nuclei_tiles(16, 256)
def measure(e, tiles, n=12):
    with torch.cuda.stream(e.stream):
        for _ in range(25): e.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize()
        hip.profile_enable(True)
        for _ in range(n): e.infer_async(tiles, hip.CH_SWAP)
        p = hip.profile_read(); hip.profile_enable(False)
    g3 = sum(v['ms'] for k, v in p.items() if k.startswith('gemm_kernel<3>')) / n
    at = sum(v['ms'] for k, v in p.items() if k.startswith('window_attn')) / n
    ml = sum(v['ms'] for k, v in p.items() if k.startswith('swin_mlp')) / n
    return round(g3, 3), round(at, 3), round(ml, 3)
out = []
a = Engine(sd, device=0, max_batch=16, tile=(256, 256))
ta = a.to_device(tiles_np)
out.append(('A', hex(a.buffer('tokens').data_ptr()), measure(a, ta)))
b = Engine(sd, device=0, max_batch=16, tile=(256, 256))
out.append(('B', hex(b.buffer('tokens').data_ptr()), measure(b, ta)))
out.append(('A again', '', measure(a, ta)))
a.close()
c = Engine(sd, device=0, max_batch=16, tile=(256, 256))
out.append(('C (after closing A)', hex(c.buffer('tokens').data_ptr()), measure(c, ta)))
print(' | '.join(f'{n} {p} g3/attn/mlp {m}' for n, p, m in out))
