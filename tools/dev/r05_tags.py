"""dev: launch time per full kernel tag (shape-resolved GEMM tags included) of the bench step, one batch at a time; schedule 0 latency / 1 throughput.
    python tools/dev/r05_tags.py [schedule=0] [steps=5]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
sched = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256), schedule=hip.SCHED_THROUGHPUT if sched else hip.SCHED_LATENCY, bind_host=True)
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
for _ in range(40): eng.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize(); hip.profile_enable(True)
for _ in range(steps): eng.infer_async(tiles, hip.CH_SWAP)
p = hip.profile_read()
tot = sum(x['ms'] for x in p.values()) / steps
print(f'schedule {sched}: {tot:.3f} ms of launches per step')
for k, x in sorted(p.items(), key=lambda kv: -kv[1]['ms']):
    ms = x['ms'] / steps
    tf = x['flops'] / (x['ms'] * 1e-3) / 1e12 if x['flops'] else 0
    print(f"{k:58s} {x['launches'] // steps:4d} launches {ms:7.3f} ms  {ms / max(1, x['launches'] // steps) * 1e3:7.1f} us each  {tf:6.1f} TFLOP/s  {x['bytes'] / (x['ms'] * 1e-3) / 1e12 if x['bytes'] else 0:5.2f} TB/s")
