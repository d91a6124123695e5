"""Micro-benchmark of the depthwise stencil kernels on the CSTS pool/upsample shapes (b=4, 16x256^2, bf16)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
lib = L.load()
# name, fine thw, C, HD, stride, count per step (k+v counted separately), fine token stride multiple (3 = inside qkv)
SH = [("b0.kv", (8, 64, 64), 96, 96, (1, 8, 8), 4), ("b1.q", (8, 64, 64), 192, 96, (1, 2, 2), 2), ("b1.kv", (8, 64, 64), 192, 96, (1, 4, 4), 4),
      ("b2.kv", (8, 32, 32), 192, 96, (1, 4, 4), 2), ("b3.q", (8, 32, 32), 384, 96, (1, 2, 2), 2), ("b3.kv", (8, 32, 32), 384, 96, (1, 2, 2), 4),
      ("b4-13.kv", (8, 16, 16), 384, 96, (1, 2, 2), 20), ("b14.q", (8, 16, 16), 768, 96, (1, 2, 2), 2), ("b14.kv", (8, 16, 16), 768, 96, (1, 1, 1), 4),
      ("b15.kv", (8, 8, 8), 768, 96, (1, 1, 1), 2),
      ("d1.q^T", (8, 16, 16), 768, 96, (1, 2, 2), 1), ("d1.kv", (8, 8, 8), 768, 96, (1, 2, 2), 2),
      ("d2.q^T", (8, 32, 32), 768, 192, (1, 2, 2), 1), ("d2.kv", (8, 16, 16), 768, 192, (1, 4, 4), 2),
      ("d3.q^T", (8, 64, 64), 384, 96, (1, 2, 2), 1), ("d3.kv", (8, 32, 32), 384, 96, (1, 8, 8), 2),
      ("d4.q^T", (16, 64, 64), 192, 96, (2, 1, 1), 1), ("d4.kv", (8, 64, 64), 192, 96, (1, 16, 16), 2)]
B = 4
def timeit(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
tot = [0, 0, 0]
print(f"{'shape':10s} {'strided':>9s} {'transp':>9s} {'wgrad':>9s}  (us)   x count")
for name, fthw, Cc, HD, st, cnt in SH:
    Nf = fthw[0] * fthw[1] * fthw[2]
    cthw = [(f - 1) // s + 1 for f, s in zip(fthw, st)]
    Nc = cthw[0] * cthw[1] * cthw[2]
    up = name.endswith("^T")
    # pools read the fine grid inside the qkv buffer (token stride 3C); upsample reads coarse inside qkv
    fine = torch.randn(B, Nf, 3 * Cc if not up else Cc, device=dev, dtype=torch.bfloat16)
    coarse = torch.randn(B, Nc, Cc if not up else 3 * Cc, device=dev, dtype=torch.bfloat16)
    w = torch.randn(HD, 27, device=dev)
    g = L.DwconvGeom()
    g.B, g.C, g.HD = B, Cc, HD
    g.Tf, g.Hf, g.Wf = fthw; g.Tc, g.Hc, g.Wc = cthw; g.st, g.sh, g.sw = st
    g.fine_batch_stride, g.fine_token_stride = fine.shape[1] * fine.shape[2], fine.shape[2]
    g.coarse_batch_stride, g.coarse_token_stride = coarse.shape[1] * coarse.shape[2], coarse.shape[2]
    s = torch.cuda.current_stream().cuda_stream
    dw = torch.empty(HD * 27, device=dev)
    ws = torch.empty(max(16, lib.csts_dwconv_wgrad_workspace(C.byref(g))), dtype=torch.uint8, device=dev)
    t1 = timeit(lambda: lib.csts_dwconv_strided(C.byref(g), fine.data_ptr(), 1, w.data_ptr(), coarse.data_ptr(), 1, s))
    t2 = timeit(lambda: lib.csts_dwconv_transposed(C.byref(g), coarse.data_ptr(), 1, w.data_ptr(), fine.data_ptr(), 1, s))
    t3 = timeit(lambda: lib.csts_dwconv_wgrad(C.byref(g), fine.data_ptr(), 1, coarse.data_ptr(), 1, dw.data_ptr(), ws.data_ptr(), ws.numel(), s))
    # per training step: pools use strided fwd + transposed bwd; upsample uses transposed fwd + strided bwd
    tot[0] += t1 * cnt; tot[1] += t2 * cnt; tot[2] += t3 * cnt
    print(f"{name:10s} {t1:9.1f} {t2:9.1f} {t3:9.1f}   x{cnt}")
print("ms/step: strided %.2f transposed %.2f wgrad %.2f" % (tot[0] / 1e3, tot[1] / 1e3, tot[2] / 1e3))
