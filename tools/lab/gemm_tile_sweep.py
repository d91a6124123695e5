"""Sweep the v2 GEMM row tile (64/128/256) and, for TN weight-gradient shapes, the split-K factor over the GEMM shapes
of one CSTS train step (read from a bench.py --dump-gemm file).  Prints us per call for every variant.
usage: gemm_tile_sweep.py <dump.txt> [top_n]"""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
rows = [l.split() for l in open(sys.argv[1]).read().strip().split("\n")[1:]]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
NSET = 3

def bench(lay, M, N, K, split, tile):
    sets = []
    for _ in range(NSET):
        if lay == "NT":   A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); lda, ldb, code = K, K, L.GEMM_NT
        elif lay == "NN": A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); lda, ldb, code = K, N, L.GEMM_NN
        else:             A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); lda, ldb, code = M, N, L.GEMM_TN
        Cm = torch.empty(M, N, device=dev, dtype=torch.float32 if lay == "TN" else torch.bfloat16)
        sets.append((A, B, Cm))
    def run(i):
        A, B, Cm = sets[i % NSET]
        ops.gemm(code, A, 0, lda, B, 0, ldb, Cm, N, M, N, K, compute=L.BF16, split_k=split, tile_rows=tile)
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 12
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

print("layout M N K calls | split:tile=us ...")
for r in rows[:top]:
    lay, M, N, K, split, calls = r[0], int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])
    if M * K > 2e8 or N * K > 2e8: continue
    splits = [split]
    if lay == "TN":
        splits = sorted({max(1, split // 4), max(1, split // 2), split, min(512, split * 2)})
    out = []
    for sp in splits:
        for tile in (64, 128, 256):
            if tile > 64 and M <= tile // 2: continue
            out.append(f"{sp}:{tile}={bench(lay, M, N, K, sp, tile):.1f}")
    print(lay, M, N, K, calls, "|", " ".join(out), flush=True)
