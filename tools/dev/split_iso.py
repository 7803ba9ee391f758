"""Dev helper: launch durations of the split-pipe GEMM on isolated shapes (events inside the library, host-side weight split excluded).
    split_iso.py [MxNxK ...]      env ISO_PIPE=fp32 for the fp32 MFMA kernel"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import weights, hip
from nuhtc_amd.engine import Engine
eng = Engine(weights.seeded_state_dict(0), device=0, max_batch=1, tile=(64, 64))
pipe = os.environ.get('ISO_PIPE', 'split')
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
out = []
for (M, N, K) in shapes or [(16384, 384, 1536), (16384, 1536, 384), (16384, 1152, 384), (65536, 768, 192), (262144, 384, 96), (262144, 96, 384), (4096, 3072, 768), (16384, 3072, 3072)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    if os.environ.get('ISO_ZERO') == '1': A.zero_(); W.zero_()
    for _ in range(2): eng.op_gemm(A, W, b, 0, pipe=pipe)
    torch.cuda.synchronize()
    hip.profile_enable(True)
    n = int(os.environ.get('ISO_N', 6))
    for _ in range(n): eng.op_gemm(A, W, b, 0, pipe=pipe)
    torch.cuda.synchronize()
    p = hip.profile_read(); hip.profile_enable(False)
    ms = sum(v['ms'] for k, v in p.items() if k.startswith('gemm')) / n
    out.append(f'M{M} N{N} K{K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF')
print(' | '.join(out))
