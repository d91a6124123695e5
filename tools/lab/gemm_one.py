import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in sys.argv[1:4]]
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16)
torch.cuda.synchronize()
