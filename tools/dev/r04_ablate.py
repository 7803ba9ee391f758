"""dev (-DNUHTC_DEV build): what each kernel group costs the step WITH FOUR BATCHES IN FLIGHT.  Fixed load (given RoIs, 64 detections per tile: the
work does not depend on the data, which goes wrong as soon as something is skipped), the engines of the throughput schedule; NUHTC_SKIP masks leave
launches out: 1 attention, 2 LayerNorm / merge-LN, 4 3x3 convolutions, 8 96-column split GEMMs, 16 fused stage-1 kernels, 32 RoI features, 64 other
GEMMs, 128 proposal chain (RPN select, NMS, watershed).  Prints ms per step in flight and one batch at a time per mask, and the difference to the full step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
size = tuple(float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '12,40').split(','))
sd = weights.bench_state_dict()
engs = [Engine(sd, device=0, max_batch=16, tile=(256, 256), schedule=hip.SCHED_THROUGHPUT) for _ in range(4)]
seq = Engine(sd, device=0, max_batch=16, tile=(256, 256))
tiles = seq.to_device(synth.nuclei_tiles(16, 256))
rois = torch.from_numpy(synth.fixed_load_rois(16, size=size)).to(tiles.device)
streams = [e.stream for e in engs]
torch.cuda.synchronize()
def run(k):
    for i in range(k):
        with torch.cuda.stream(streams[i % 4]):
            engs[i % 4].infer_fixed_load_async(tiles, rois, 64, hip.CH_SWAP)
def timed(fn, k):
    fn(8); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(k); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
def run_seq(k):
    with torch.cuda.stream(seq.stream):
        for _ in range(k): seq.infer_fixed_load_async(tiles, rois, 64, hip.CH_SWAP)
for _ in range(3): timed(run, 40)
NAMES = {0: 'full step', 1: 'attention', 2: 'LayerNorm / merge-LN', 4: '3x3 convolutions', 8: '96-column split GEMMs', 16: 'fused stage-1 kernels', 32: 'RoI features',
         64: 'other GEMMs (FCs, 1x1, deconv, merges)', 128: 'proposal chain', 1 | 2 | 32 | 128: 'attention + LN + RoI features + proposals (everything that is not a matrix kernel)',
         4 | 8 | 16 | 64: 'every matrix kernel'}
res = {}
for rnd in range(3):
    for m in NAMES:
        hip.dev_knob('SKIP', m)
        res.setdefault(m, []).append((timed(run, 40), timed(run_seq, 20)))
hip.dev_knob('SKIP', 0)
f0 = np.median([a for a, _ in res[0]]); s0 = np.median([b for _, b in res[0]])
print(f'fixed load, RoI sides {size}: in flight / one at a time, ms per step (median of 3)')
for m, nm in NAMES.items():
    f = np.median([a for a, _ in res[m]]); s = np.median([b for _, b in res[m]])
    print(f'  without {nm:75s} {f:7.3f} ({f0 - f:+.3f})   {s:7.3f} ({s0 - s:+.3f})' if m else f'  {nm:83s} {f:7.3f}            {s:7.3f}')
