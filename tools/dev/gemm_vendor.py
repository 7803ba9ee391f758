"""Dev helper: the vendor fp32 GEMM (torch.nn.functional.linear -> rocBLAS / hipBLASLt) on the Swin shapes, beside nuhtc_op_gemm."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import weights
from nuhtc_amd.engine import Engine
eng = Engine(weights.seeded_state_dict(0), device=0, max_batch=1, tile=(64, 64))
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(262144, 384, 96), (262144, 288, 96), (262144, 96, 384), (65536, 768, 192), (65536, 192, 768), (16384, 1536, 384), (16384, 384, 1536), (4096, 3072, 768), (16384, 3072, 3072)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in shapes:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
    if os.environ.get('ISO_ZERO') == '1': A.zero_(); W.zero_()
    t_own = timeit(lambda: eng.op_gemm(A, W, b, 0))
    t_ven = timeit(lambda: torch.nn.functional.linear(A, W, b))
    fl = 2.0 * M * N * K / 1e9
    print(f'M{M} N{N} K{K}: nuhtc {t_own:.3f} ms {fl/t_own:.1f} TF | vendor {t_ven:.3f} ms {fl/t_ven:.1f} TF')
