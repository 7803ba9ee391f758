"""dev: is the fast / slow state of the dense launches fixed per PROCESS or per set of allocations?  One process creates an engine, measures the
96-column GEMM time per step, destroys it, allocates (and keeps or frees) filler memory of a random size, creates the next engine ... N times;
then measures several engines that are alive together.   usage: r04_state_reroll.py [N] [seed]"""
import gc, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else os.getpid())
sd = weights.bench_state_dict()
tiles_h = synth.nuclei_tiles(16, 256)
def measure(eng, tiles, warm=30):
    torch.cuda.set_stream(eng.stream)
    for _ in range(warm): eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize()
    hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    return sum(v['ms'] for k, v in p.items() if k.startswith('gemm_kernel<3>')) / 5
out = []
fill = []
for r in range(N):
    eng = Engine(sd, device=0, max_batch=16, tile=(256, 256))
    tiles = eng.to_device(tiles_h)
    g = measure(eng, tiles)
    out.append(g)
    print(f'engine {r}: gemm3 {g:.3f} ms  (filler held: {sum(f.numel() for f in fill) >> 20} MiB)', flush=True)
    eng.close(); del eng, tiles; gc.collect(); torch.cuda.empty_cache()
    # shift what the next engine gets: hold a filler of 100 MiB .. 3 GiB (odd multiples of 2 MiB), sometimes drop the older ones
    if rnd.random() < 0.4: fill.clear(); torch.cuda.empty_cache()
    fill.append(torch.empty(((rnd.randrange(50, 1500) * 2 + 1) << 20), dtype=torch.uint8, device='cuda'))
print('sequential engines:', ' '.join(f'{g:.3f}' for g in out))
fill.clear(); torch.cuda.empty_cache()
engs = [Engine(sd, device=0, max_batch=16, tile=(256, 256)) for _ in range(4)]
both = []
for e in engs:
    t = e.to_device(tiles_h)
    both.append(measure(e, t))
print('four engines alive together, measured one at a time:', ' '.join(f'{g:.3f}' for g in both))
