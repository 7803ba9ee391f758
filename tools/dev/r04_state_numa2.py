"""dev: gemm_kernel<3> per step and the sequential step of one engine; prints this process's GPU (PCI id, NUMA node), the CPUs it may run on, and --
with MOVE=<cpulist> -- measures again after moving the process there (is it where the process STARTED or where it RUNS?)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
def cpus(s):
    out = set()
    for part in s.split(','):
        a, _, b = part.partition('-'); out |= set(range(int(a), int(b or a) + 1))
    return out
bdf = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), 'pci_bus_id') else '?'
node = '?'
for d in os.listdir('/sys/bus/pci/devices'):
    if bdf != '?' and d.lower().endswith(str(bdf).lower()[-7:]):
        try: node = open(f'/sys/bus/pci/devices/{d}/numa_node').read().strip(); bdf = d
        except OSError: pass
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
def measure(warm=30):
    for _ in range(warm): eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): eng.infer_async(tiles, hip.CH_SWAP)
    torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 30 * 1e3
    hip.profile_enable(True)
    for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
    p = hip.profile_read(); hip.profile_enable(False)
    return sum(v['ms'] for k, v in p.items() if k.startswith('gemm_kernel<3>')) / 5, step
g, st = measure()
aff = sorted(os.sched_getaffinity(0))
print(f'gpu {bdf} numa_node {node} | cpus {aff[0]}..{aff[-1]} ({len(aff)}) | HIP_FORCE_DEV_KERNARG={os.environ.get("HIP_FORCE_DEV_KERNARG")} | gemm3 {g:.3f} ms, step {st:.3f} ms', flush=True)
if os.environ.get('MOVE'):
    os.sched_setaffinity(0, cpus(os.environ['MOVE'])); time.sleep(0.2)
    g, st = measure()
    print(f'   moved to {os.environ["MOVE"]}: gemm3 {g:.3f} ms, step {st:.3f} ms (same engine)', flush=True)
    eng2 = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
    eng = eng2; torch.cuda.set_stream(eng.stream); tiles = eng.to_device(synth.nuclei_tiles(16, 256))
    g, st = measure()
    print(f'   a new engine created there: gemm3 {g:.3f} ms, step {st:.3f} ms', flush=True)
