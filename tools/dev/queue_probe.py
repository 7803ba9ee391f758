"""Dev probe: which stream placements (GPU_MAX_HW_QUEUES > 4) are slow relative to a main stream?  Streams are created raw
(hipStreamCreateWithFlags, creation order = index) and wrapped as torch ExternalStreams; for each (main, candidate) pair:
 conc: two 200-us spin kernels, one per stream, started together -> elapsed (200 = concurrent, 400 = serialized)
 ping: 50 fork/join round trips of tiny kernels (main -> cand -> main) -> us per round trip"""
import ctypes, os, sys, time, torch
hipl = ctypes.CDLL('libamdhip64.so')
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
def mk():
    s = ctypes.c_void_p()
    assert hipl.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return torch.cuda.ExternalStream(s.value)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
main_idx = int(sys.argv[2]) if len(sys.argv) > 2 else -1      # -1: the null stream
streams = [mk() for _ in range(N)]
x = torch.zeros(64, device='cuda')
for s in streams:                      # touch every stream once (queues are bound lazily)
    with torch.cuda.stream(s): x.add_(0)
torch.cuda.synchronize()
main = torch.cuda.default_stream() if main_idx < 0 else streams[main_idx]
spin = int(200e-6 * 100e6 * 21)        # torch.cuda._sleep counts shader cycles roughly; calibrated below
def t_sleep(cyc):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(main): torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
t1 = t_sleep(spin); spin = int(spin * 200.0 / max(t1, 1.0)); base = t_sleep(spin)
res = []
for i, c in enumerate(streams):
    if c is main: res.append((i, None, None)); continue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(main): torch.cuda._sleep(spin)
    with torch.cuda.stream(c): torch.cuda._sleep(spin)
    torch.cuda.synchronize(); conc = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        with torch.cuda.stream(main): x.add_(1)
        c.wait_stream(main)
        with torch.cuda.stream(c): x.add_(1)
        main.wait_stream(c)
    torch.cuda.synchronize(); ping = (time.perf_counter() - t0) * 1e6 / 50
    res.append((i, round(conc), round(ping, 1)))
print('queues', os.environ.get('GPU_MAX_HW_QUEUES', 'default'), 'main', main_idx, 'one spin', round(base), 'us; (stream, conc us, ping us):')
print('  ', res)
