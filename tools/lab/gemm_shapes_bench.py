"""Times every GEMM shape of one training step (from a bench.py --dump-gemm table) in isolation."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
rows = [l.split() for l in open(sys.argv[1]).read().strip().split("\n")[1:]]
lay = {"NT": L.GEMM_NT, "NN": L.GEMM_NN, "TN": L.GEMM_TN}
out = []
for lo, M, N, K, sp, calls, *_ in rows:
    M, N, K, sp, calls = int(M), int(N), int(K), int(sp), int(calls) // 2
    if lo == "NT":
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); lda, ldb = K, K
    elif lo == "NN":
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); lda, ldb = K, N
    else:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); lda, ldb = M, N
        sp = ops._wgrad_split(M, N, K)
    Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if lo == "TN" else torch.bfloat16)
    f = lambda: ops.gemm(lay[lo], A, 0, lda, B, 0, ldb, Cm, N, M, N, K, compute=L.BF16, split_k=sp)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    byt = A.numel() * A.element_size() + B.numel() * B.element_size() + Cm.numel() * Cm.element_size()
    ideal = max(2.0 * M * N * K / 1.5e15, byt / 5e12) * 1e6
    out.append((us * calls, lo, M, N, K, sp, calls, us, 2.0 * M * N * K / us / 1e6, ideal, byt / us / 1e3))
out.sort(reverse=True)
tot = sum(o[0] for o in out)
print(f"total GEMM ms/step {tot/1e3:.2f}")
print(f"ideal (1.5 PF/s, 5 TB/s) total ms/step {sum(o[9]*o[6] for o in out)/1e3:.2f}")
for t, lo, M, N, K, sp, calls, us, tf, ideal, gbs in out[:60]:
    print(f"{lo} {M:7d} {N:6d} {K:7d} split {sp:3d} calls {calls:3d} {us:8.1f} us {tf:7.1f} TF/s {gbs:6.0f} GB/s ideal {ideal:6.1f} us x{us/ideal:4.1f} -> {t/1e3:6.2f} ms/step (gap {(us-ideal)*calls/1e3:5.2f})")
