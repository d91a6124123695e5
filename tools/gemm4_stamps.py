"""Shader-clock stamps of the 8-wave LDS-DMA NT kernel (diagnostics build: `make -C csts_amd/csrc stamps` ->
tools/diag/libcsts_hip_stamps.so).  For workgroups 0 and 137, per output tile: cycles of every k-step as wait (own LDS-DMA
share) / barrier / fragment reads + MFMAs + refill issue, then the epilogue (until its stores are ISSUED), and the final drain.
usage: gemm4_stamps.py M N K algo [none|gelu|res]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libcsts_hip_stamps.so")
from csts_amd import ops
dev = torch.device("cuda:0")
M, N, K, algo = [int(v) for v in sys.argv[1:5]]
epi = sys.argv[5] if len(sys.argv) > 5 else "none"
A = torch.randn(M, K, device=dev).bfloat16(); B = (0.1 * torch.randn(N, K, device=dev)).bfloat16()
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev)
aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == "gelu" else None
buf = torch.zeros(1024, device=dev, dtype=torch.int64)
other = torch.randn(64 << 20, device=dev)          # 256 MB: evict the operands from L2 / Infinity Cache between runs
for it in range(3):
    other.add_(1.0)
    torch.cuda.synchronize()
    if epi == "gelu":
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, bias=bias, epilogue=L.EPI_GELU, aux=aux, algo=algo, debug_ws=buf)
    else:
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, bias=bias, algo=algo, debug_ws=buf)
    torch.cuda.synchronize()
nk = (K + 63) // 64
for w in range(2):
    s = buf[w * 512:(w + 1) * 512].tolist()
    n = s[0]
    t = s[1:n]
    if len(t) < 4: continue
    real = (s[501] - s[500]) / 100.0      # us (100 MHz)
    print(f"workgroup {w}: total {t[-1] - t[0]} cycles = {real:.1f} us -> {(t[-1] - t[0]) / max(real, 1e-9) / 1e3:.2f} GHz; prologue issue {t[1] - t[0]}; final drain {t[-1] - t[-2]}")
    i = 2
    tile = 0
    while i + 3 * nk + 2 <= len(t) - 1 + 1:
        rows = []
        for kt in range(nk):
            a, b, c = t[i:i + 3]
            prev = t[i - 1]
            rows.append((a - prev, b - a, c - b))
            i += 3
        if i + 1 >= len(t): break
        e0, e1 = t[i], t[i + 1]
        i += 2
        ks = sum(sum(r) for r in rows)
        print(f"  tile {tile}: k-loop {ks} (" + " ".join(f"{r[0]}/{r[1]}/{r[2]}" for r in rows[:8]) + (" ..." if nk > 8 else "") + f") | epilogue {e1 - e0}")
        tile += 1
