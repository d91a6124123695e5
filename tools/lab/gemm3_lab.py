"""A/B of the NT GEMM kernels (register-staged gemm2 vs the LDS-DMA ring gemm3 variants) on the NT shapes of one CSTS
train step: correctness against torch (fp32 reference of the bf16 operands) and interleaved timings in one process.
usage: gemm3_lab.py [shapes-file]   (lines 'M N K'; default = the hot NT shapes of b=4, 16x256^2)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
SHAPES = [(8192, 1536, 384), (8192, 384, 1536), (8192, 1152, 384), (8192, 384, 384), (131072, 192, 384), (131072, 576, 192),
          (131072, 768, 384), (262144, 384, 192), (131072, 384, 96), (32768, 1152, 384), (32768, 768, 192), (8192, 2304, 768),
          (32768, 1536, 768), (32768, 384, 768), (2048, 768, 3072), (2048, 3072, 768), (262144, 192, 192), (262144, 96, 384),
          (131072, 288, 96), (2080, 768, 3072), (131072, 96, 160)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in l.split()[:3]) for l in open(sys.argv[1]) if l.strip()]
ALGOS = [0, 312, 313, 322, 1322, 323, 342]
NSET, ROUNDS, REP = 3, 3, 8

def run(algo, A, B, Cm, M, N, K, bias=None):
    ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, bias=bias, algo=algo)

print("M N K | " + " ".join(f"a{a}" for a in ALGOS) + "   (us per call, min over rounds; TF/s of the best)")
for (M, N, K) in SHAPES:
    sets = [(torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16(),
             torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(NSET)]
    bias = torch.randn(N, device=dev)
    A, B, Cm = sets[0]
    rows = torch.randint(0, M, (256,), device=dev)
    ref = A[rows].float() @ B.float().t() + bias
    errs = []
    for a in ALGOS:
        Cm.zero_()
        run(a, A, B, Cm, M, N, K, bias)
        torch.cuda.synchronize()
        e = ((Cm[rows].float() - ref).norm() / ref.norm()).item()
        errs.append(e)
    best = {a: 1e9 for a in ALGOS}
    for r in range(ROUNDS):
        for a in ALGOS:
            run(a, *sets[0], M, N, K)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(REP): run(a, *sets[i % NSET], M, N, K)
            e1.record(); torch.cuda.synchronize()
            best[a] = min(best[a], e0.elapsed_time(e1) * 1e3 / REP)
    bst = min(best.values())
    flag = "" if max(errs) < 6e-3 else "  ** ERR " + " ".join(f"{e:.1e}" for e in errs)
    print(M, N, K, "|", " ".join(f"{best[a]:.1f}" for a in ALGOS), f"  best {2.0 * M * N * K / bst / 1e6:.0f} TF/s vs a0 {2.0 * M * N * K / best[0] / 1e6:.0f}" + flag, flush=True)
