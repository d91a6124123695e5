"""Follow-up to gemm_sustained.py: what makes a GEMM of the train step take ~1.5x its back-to-back time?  One shape
(8192 x 1536 x 384, fc1 of the 384-channel stage), library default kernel, event pair per call:
 (1) bias only, 24 operand sets;  (2) GELU + pre-activation output (the real fc1 epilogue), 24 sets;
 (3) as (2) over 160 operand sets (~9 GB: the footprint a train step walks through);
 (4) as (2), each call preceded by a DIFFERENT kernel (a LayerNorm over the same rows), as in the step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
M, N, K = 8192, 1536, 384
def mk(n):
    return [(torch.randn(M, K, device=dev).bfloat16(), (0.1 * torch.randn(N, K, device=dev)).bfloat16(),
             torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(n)]
bias = torch.randn(N, device=dev)
def run(s, gelu):
    A, B, C_, aux = s
    if gelu:
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, C_, N, M, N, K, compute=L.BF16, bias=bias, epilogue=L.EPI_GELU, aux=aux)
    else:
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, C_, N, M, N, K, compute=L.BF16, bias=bias)
def timed(sets, n, gelu, pre=None):
    evs = []
    for i in range(n):
        if pre is not None:
            pre(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(sets[i % len(sets)], gelu); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[n // 4:])
    return t[len(t) // 2], t[len(t) // 10], t[9 * len(t) // 10]
sets = mk(24)
for _ in range(3): run(sets[0], True)
torch.cuda.synchronize()
print("(1) bias only, 24 sets            : median %.1f us (p10 %.1f, p90 %.1f)" % timed(sets, 1200, False))
print("(2) GELU + aux, 24 sets           : median %.1f us (p10 %.1f, p90 %.1f)" % timed(sets, 1200, True))
big = sets + mk(136)
print("(3) GELU + aux, 160 sets (~9 GB)  : median %.1f us (p10 %.1f, p90 %.1f)" % timed(big, 1200, True))
x = torch.randn(M, K, device=dev); g = torch.ones(K, device=dev); b = torch.zeros(K, device=dev)
def ln(i):
    ops.layer_norm(x, g, b, 1e-6, L.BF16)
print("(4) GELU + aux, 24 sets, LN before: median %.1f us (p10 %.1f, p90 %.1f)" % timed(sets, 1200, True, ln))
print("(5) GELU + aux, 160 sets, LN before: median %.1f us (p10 %.1f, p90 %.1f)" % timed(big, 1200, True, ln))
