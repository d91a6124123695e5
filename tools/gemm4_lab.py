"""A/B of the NT GEMM kernels on the NT shapes of one CSTS train step: the library default (register-staged gemm2 / LDS-DMA
ring gemm3) against the 8-wave LDS-DMA ring variants of gemm4.hip -- correctness against torch (fp32 reference of the bf16
operands, with the bias / GELU / residual epilogues) and interleaved timings in one process, rotating over NSET operand
sets so that consecutive launches do not find their operands in L2.
usage: gemm4_lab.py [--shapes file] [--algos 402,403,...] [--epi none|gelu|res]   (shape lines 'M N K')"""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
SHAPES = [(8192, 1536, 384), (8192, 384, 1536), (8192, 1152, 384), (8192, 384, 384), (131072, 192, 384), (131072, 576, 192),
          (131072, 768, 384), (262144, 384, 192), (32768, 1152, 384), (32768, 768, 192), (8192, 2304, 768),
          (32768, 1536, 768), (32768, 384, 768), (2048, 768, 3072), (2048, 3072, 768), (262144, 192, 192), (262144, 96, 384),
          (131072, 384, 384), (131072, 192, 768), (32768, 768, 768), (8192, 768, 1536), (8192, 3072, 768), (8192, 768, 3072)]
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default=None)
ap.add_argument("--algos", default="0,402,403,412,422,432,433,434,462,463,442,452")
ap.add_argument("--epi", default="none", choices=["none", "gelu", "res", "dgelu"])
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
if a.shapes:
    SHAPES = [tuple(int(v) for v in l.split()[:3]) for l in open(a.shapes) if l.strip() and not l.startswith("#")]
ALGOS = [int(v) for v in a.algos.split(",")]
NSET, ROUNDS, REP = 3, a.rounds, 8


def run(algo, A, B, Cm, M, N, K, bias, aux, res):
    if algo < 0:            # the vendor library through torch (hipBLASLt / rocBLAS), bias epilogue only
        torch.addmm(bias16, A, B.t(), out=Cm) if Cm.dtype == torch.bfloat16 else torch.mm(A, B.t())
        return
    if a.epi == "gelu":
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, bias=bias, epilogue=L.EPI_GELU, aux=aux, algo=algo)
    elif a.epi == "res":
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, bias=bias, residual=res, ldr=N, algo=algo)
    elif a.epi == "dgelu":        # the data gradient through GELU: x gelu'(saved pre-activation), no bias (aux is READ here)
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, epilogue=L.EPI_DGELU, aux=aux, algo=algo)
    else:
        ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16, bias=bias, algo=algo)


print(f"epilogue={a.epi}   M N K | " + " ".join(f"a{x}" for x in ALGOS) + "   (us per call, min over rounds)")
for (M, N, K) in SHAPES:
    cdt = torch.float32 if a.epi == "res" else torch.bfloat16
    sets = [(torch.randn(M, K, device=dev).bfloat16(), (0.1 * torch.randn(N, K, device=dev)).bfloat16(),
             torch.empty(M, N, device=dev, dtype=cdt)) for _ in range(NSET)]
    bias = torch.randn(N, device=dev)
    bias16 = bias.bfloat16()
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if a.epi == "gelu" else (torch.randn(M, N, device=dev).bfloat16() if a.epi == "dgelu" else None)
    res = torch.randn(M, N, device=dev) if a.epi == "res" else None
    A, B, Cm = sets[0]
    rows = torch.cat([torch.randint(0, M, (192,), device=dev), torch.arange(M - 64, M, device=dev)])
    pre = A[rows].float() @ B.float().t() + bias
    ref = F.gelu(pre) if a.epi == "gelu" else (pre + res[rows] if a.epi == "res" else pre)
    if a.epi == "dgelu":
        hh = aux[rows].float().requires_grad_(True)
        F.gelu(hh).sum().backward()
        ref = (pre - bias) * hh.grad
    errs = []
    for al in ALGOS:
        Cm.zero_()
        if aux is not None and a.epi == "gelu":
            aux.zero_()
        try:
            run(al, A, B, Cm, M, N, K, bias, aux, res)
            torch.cuda.synchronize()
            e = ((Cm[rows].float() - ref).norm() / ref.norm()).item()
            if aux is not None and a.epi == "gelu":
                e = max(e, ((aux[rows].float() - pre).norm() / pre.norm()).item())
        except L.CstsError as ex:
            e = float("nan")
        errs.append(e)
    best = {al: float("inf") for al in ALGOS}
    for r in range(ROUNDS):
        for al, e in zip(ALGOS, errs):
            if e != e:
                continue
            run(al, *sets[0], M, N, K, bias, aux, res)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(REP):
                run(al, *sets[i % NSET], M, N, K, bias, aux, res)
            e1.record(); torch.cuda.synchronize()
            best[al] = min(best[al], e0.elapsed_time(e1) * 1e3 / REP)
    bst = min(best.values())
    who = min(best, key=best.get)
    bad = [f"a{al}:{e:.1e}" for al, e in zip(ALGOS, errs) if e == e and e > 6e-3]
    flag = ("  ** ERR " + " ".join(bad)) if bad else ""
    print(M, N, K, "|", " ".join(("  n/a" if best[al] == float("inf") else f"{best[al]:.1f}") for al in ALGOS),
          f"  best a{who} {2.0 * M * N * K / bst / 1e6:.0f} TF/s vs a0 {2.0 * M * N * K / best[ALGOS[0]] / 1e6:.0f} ({best[ALGOS[0]] / bst:.2f}x)" + flag, flush=True)
