"""Dev helper: time nuhtc_op_gemm on isolated shapes."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import weights
from nuhtc_amd.engine import Engine
eng = Engine(weights.seeded_state_dict(0), device=0, max_batch=1, tile=(64, 64))
ZERO = os.environ.get('ISO_ZERO') == '1'   # zero operands: same instruction stream, minimal switching power
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (M, N, K) in shapes or [(16384, 3072, 3072), (8192, 3072, 768), (262144, 384, 96), (262144, 96, 384), (283024, 288, 96), (16384, 1536, 384), (4096, 768, 3072), (65536, 768, 192)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    if ZERO: A.zero_(); W.zero_()
    for _ in range(3): eng.op_gemm(A, W, b, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n): eng.op_gemm(A, W, b, 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'M{M} N{N} K{K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.1f} TF   {4.0*(M*K+N*K+M*N)/ms/1e6:.0f} GB/s')
