import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from nuhtc_amd import weights
from nuhtc_amd.engine import Engine
eng = Engine(weights.seeded_state_dict(0), device=0, max_batch=1, tile=(64, 64))
M, N, K = 16384, 3072, 3072
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
for _ in range(12): eng.op_gemm(A, W, b, 0)
torch.cuda.synchronize()
