"""Yardstick for the GEMM family: every GEMM shape of one training step (bench.py --dump-gemm table) timed in
isolation with csts_gemm and, as an on-box cross-check only (never on the product path), with the vendor library
behind torch.matmul.  Buffers rotate over several copies so that operands are not L2-warm from the previous call.

usage: python tools/gemm_yardstick.py profiles/r1_v3_gemm_shapes.txt [max_rows]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L

dev = torch.device("cuda:0")
rows = [l.split() for l in open(sys.argv[1]).read().strip().split("\n")[1:]]
if len(sys.argv) > 2:
    rows = rows[:int(sys.argv[2])]
lay = {"NT": L.GEMM_NT, "NN": L.GEMM_NN, "TN": L.GEMM_TN}
NB = 4
REPS = 8


def timeit(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (REPS * len(fns)) * 1e3


out = []
for lo, M, N, K, sp, calls, *_ in rows:
    M, N, K, sp, calls = int(M), int(N), int(K), int(sp), int(calls) // 2
    mine, ven = [], []
    for i in range(NB):
        if lo == "NT":
            A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); lda, ldb = K, K
            v = (lambda A=A, B=B: torch.matmul(A, B.t()))
        elif lo == "NN":
            A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); lda, ldb = K, N
            v = (lambda A=A, B=B: torch.matmul(A, B))
        else:
            A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); lda, ldb = M, N
            sp = ops._wgrad_split(M, N, K)
            v = (lambda A=A, B=B: torch.matmul(A.t(), B))
        Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if lo == "TN" else torch.bfloat16)
        mine.append(lambda A=A, B=B, Cm=Cm, lda=lda, ldb=ldb, sp=sp: ops.gemm(lay[lo], A, 0, lda, B, 0, ldb, Cm, N, M, N, K,
                                                                              compute=L.BF16, split_k=sp))
        ven.append(v)
    us_m, us_v = timeit(mine), timeit(ven)
    out.append((us_m * calls, lo, M, N, K, sp, calls, us_m, us_v))
    del mine, ven
out.sort(reverse=True)
print(f"total ms/step: csts_gemm {sum(o[0] for o in out)/1e3:.2f}   vendor (torch.matmul, bf16 out) {sum(o[8]*o[6] for o in out)/1e3:.2f}")
for t, lo, M, N, K, sp, calls, us_m, us_v in out:
    fl = 2.0 * M * N * K
    print(f"{lo} {M:7d} {N:6d} {K:7d} split {sp:3d} x{calls:3d}  ours {us_m:8.1f} us {fl/us_m/1e6:7.1f} TF/s | vendor {us_v:8.1f} us {fl/us_v/1e6:7.1f} TF/s | ratio {us_m/us_v:5.2f}")
