"""Dev helper: A/B settings of the launch heuristics inside ONE process, alternating every few steps so that clock drift of the box
hits every setting alike.   knob_ab.py NAME v0 v1 [v2 ...] [--rounds 12] [--steps 10]
Prints ms per step per setting (mean and spread over the rounds) and the GEMM tag times of the last round."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
ap = argparse.ArgumentParser()
ap.add_argument('name'); ap.add_argument('values', nargs='+', type=int)
ap.add_argument('--rounds', type=int, default=12); ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--tags', default='gemm', help='comma-separated tag prefixes whose launch times are summed per setting')
args = ap.parse_args()
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
for _ in range(40): eng.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize()
t = {v: [] for v in args.values}
for r in range(args.rounds):
    for v in (args.values if r % 2 == 0 else args.values[::-1]):
        hip.dev_knob(args.name, v)
        for _ in range(2): eng.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps): eng.infer_async(tiles, hip.CH_SWAP)
        torch.cuda.synchronize(); t[v].append((time.perf_counter() - t0) / args.steps * 1e3)
for v in args.values:
    a = np.array(t[v]); print(f'{args.name}={v}: {a.mean():.3f} ms per step (median {np.median(a):.3f}, min {a.min():.3f}, max {a.max():.3f}) -> {16e3 / a.mean():.1f} tiles/s')
for v in args.values:
    hip.dev_knob(args.name, v)
    hip.profile_enable(True)
    for _ in range(3): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    g = {}
    for k, x in p.items():
        if any(k.startswith(t) for t in args.tags.split(',')): g[k.split('|')[0]] = g.get(k.split('|')[0], 0) + x['ms'] / 3
    print(f'  {args.name}={v}: tags', {k: round(x, 3) for k, x in sorted(g.items())}, 'sum', round(sum(g.values()), 3))
