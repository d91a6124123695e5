"""Cycle stamps of the persistent NT kernel (diagnostics build: gemm.hip compiled with -DCSTS_GEMM3_STAMPS and linked as
tools/diag/libcsts_hip_stamps.so by `make -C csts_amd/csrc stamps`).  Prints, for two workgroups, the cycles between the stamp points of every k-step:
wait (vmcnt) | barrier | fragment reads + MFMAs with the interleaved LDS-DMA refill; and the epilogue.
usage: gemm3_stamps.py M N K algo"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libcsts_hip_stamps.so")
from csts_amd import ops
dev = torch.device("cuda:0")
M, N, K, algo = [int(v) for v in sys.argv[1:5]]
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
buf = torch.zeros(1024, device=dev, dtype=torch.int64)
for _ in range(3):
    ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, algo=algo, debug_ws=buf)
torch.cuda.synchronize()
nk = (K + 63) // 64
for w in range(2):
    s = buf[w * 512:(w + 1) * 512].tolist()
    n = s[0]
    t = s[1:n]
    if not t: continue
    print(f"workgroup {w}: {n - 1} stamps, total {t[-1] - t[0]} cycles; prologue issue {t[1] - t[0]}")
    i = 2
    tile = 0
    while i + 3 * nk + 2 <= len(t):
        rows = []
        for kt in range(nk):
            a, b, c = t[i:i + 3]
            prev = t[i - 1]
            rows.append((a - prev, b - a, c - b))
            i += 3
        epi0, epi1 = t[i], t[i + 1]
        i += 2
        print(f"  tile {tile}: k-steps (wait,barrier,mfma+refill): " + " ".join(f"{r[0]}/{r[1]}/{r[2]}" for r in rows) + f" | epilogue {epi1 - epi0}")
        tile += 1
