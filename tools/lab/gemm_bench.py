"""Micro-benchmark of csts_gemm on the shapes of the CSTS training step (b=4, 16x256^2)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L

dev = torch.device("cuda:0")
SHAPES = [  # layout, M, N, K, a_dt, b_dt, c_dt, split
    ("NT", 8192, 384, 384, "bf16", "f32", "f32", 1),
    ("NT", 8192, 1536, 384, "bf16", "f32", "bf16", 1),
    ("NT", 8192, 384, 1536, "bf16", "f32", "f32", 1),
    ("NT", 8192, 1152, 384, "bf16", "f32", "bf16", 1),
    ("NT", 131072, 576, 192, "bf16", "f32", "bf16", 1),
    ("NT", 262144, 384, 96, "bf16", "f32", "bf16", 1),
    ("NT", 2048, 768, 3072, "bf16", "f32", "f32", 1),
    ("NT", 32768, 1536, 768, "bf16", "f32", "bf16", 1),
    ("NT", 8192, 2304, 768, "bf16", "f32", "bf16", 1),
    ("NN", 8192, 384, 1536, "f32", "f32", "bf16", 1),
    ("NN", 8192, 1536, 384, "bf16", "f32", "bf16", 1),
    ("NN", 131072, 384, 192, "f32", "f32", "bf16", 1),
    ("NN", 32768, 384, 1152, "bf16", "f32", "bf16", 1),
    ("TN", 1536, 384, 8192, "bf16", "bf16", "f32", 28),
    ("TN", 384, 1536, 8192, "f32", "bf16", "f32", 28),
    ("TN", 576, 192, 131072, "bf16", "bf16", "f32", 102),
    ("TN", 768, 49152, 32, "f32", "bf16", "f32", 1),
    ("NT", 32, 768, 49152, "bf16", "f32", "f32", 128),
    ("NT", 4096, 4096, 4096, "bf16", "bf16", "bf16", 1),
    ("NT", 8192, 8192, 8192, "bf16", "bf16", "bf16", 1),
]
td = {"bf16": torch.bfloat16, "f32": torch.float32}
lay = {"NT": L.GEMM_NT, "NN": L.GEMM_NN, "TN": L.GEMM_TN}
print(f"{'layout':6s} {'M':>7s} {'N':>6s} {'K':>7s} split   us     TFLOP/s  GB/s")
for (lo, M, N, K, ad, bd, cd, split) in SHAPES:
    if lo == "NT":
        A = torch.randn(M, K, device=dev).to(td[ad]); B = torch.randn(N, K, device=dev).to(td[bd]); lda, ldb = K, K
    elif lo == "NN":
        A = torch.randn(M, K, device=dev).to(td[ad]); B = torch.randn(K, N, device=dev).to(td[bd]); lda, ldb = K, N
    else:
        A = torch.randn(K, M, device=dev).to(td[ad]); B = torch.randn(K, N, device=dev).to(td[bd]); lda, ldb = M, N
    Cm = torch.zeros(M, N, device=dev, dtype=td[cd])
    def run():
        ops.gemm(lay[lo], A, 0, lda, B, 0, ldb, Cm, N, M, N, K, compute=L.BF16, split_k=split)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    by = A.numel() * A.element_size() + B.numel() * B.element_size() + Cm.numel() * Cm.element_size()
    print(f"{lo:6s} {M:7d} {N:6d} {K:7d} {split:4d} {us:8.1f} {2.0*M*N*K/us/1e6:8.1f} {by/us/1e3:7.0f}")
