"""dev: is the fast / slow state of the 96-column GEMMs tied to the STREAM (hardware queue) the engine runs on?  One engine; gemm_kernel<3> per step
measured on its own stream, the null stream and S fresh streams, each visited ROUNDS times.   usage: r04_state_stream.py [S] [ROUNDS]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
torch.cuda.synchronize()
def measure(warm=12):
    for _ in range(warm): eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize()
    hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    return sum(v['ms'] for k, v in p.items() if k.startswith('gemm_kernel<3>')) / 5
streams = [('engine', eng.stream), ('null', torch.cuda.default_stream())] + [(f's{i}', torch.cuda.Stream()) for i in range(S)]
res = {n: [] for n, _ in streams}
for r in range(ROUNDS):
    for n, s in streams:
        with torch.cuda.stream(s):
            res[n].append(measure(30 if r == 0 else 12))
for n, s in streams:
    print(f'{n:7s} {s.cuda_stream:#x}: ' + ' '.join(f'{g:.3f}' for g in res[n]))
