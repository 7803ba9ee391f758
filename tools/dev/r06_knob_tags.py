"""dev (round 6): A/B of a dev knob on ONE full kernel tag (shape-resolved), fixed load (the RoI / detection counts do not follow the results,
so knobs that alter results still time the same work), alternating in one process.
    python tools/dev/r06_knob_tags.py NAME v0 v1 ... --tag 'gemm_kernel<2>|N256|K3136' [--roi-size 12,40] [--rounds 8]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
ap = argparse.ArgumentParser()
ap.add_argument('name'); ap.add_argument('values', nargs='+', type=int)
ap.add_argument('--tag', required=True); ap.add_argument('--rounds', type=int, default=8); ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--roi-size', default='12,40')
args = ap.parse_args()
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256), bind_host=True)
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
rois = torch.from_numpy(synth.fixed_load_rois(16, size=tuple(float(v) for v in args.roi_size.split(',')))).to(tiles.device)
step = lambda: eng.infer_fixed_load_async(tiles, rois, 64, hip.CH_SWAP)
for _ in range(40): step()
torch.cuda.synchronize()
t = {v: [] for v in args.values}
for r in range(args.rounds):
    for v in (args.values if r % 2 == 0 else args.values[::-1]):
        hip.dev_knob(args.name, v)
        for _ in range(2): step()
        torch.cuda.synchronize(); hip.profile_enable(True)
        for _ in range(args.steps): step()
        p = hip.profile_read(); hip.profile_enable(False)
        sel = [x for k, x in p.items() if k == args.tag or (args.tag.endswith('*') and k.startswith(args.tag[:-1]))]
        t[v].append((sum(x['ms'] for x in sel) / max(1, sum(x['launches'] for x in sel)) * 1e3, sum(x['flops'] for x in sel) / max(1e-9, sum(x['ms'] for x in sel)) / 1e9))
for v in args.values:
    a = np.array(t[v])
    print(f'{args.name}={v}: {args.tag}: {a[:, 0].mean():.1f} us per launch (min {a[:, 0].min():.1f}, max {a[:, 0].max():.1f}), {a[:, 1].mean():.1f} TFLOP/s')
