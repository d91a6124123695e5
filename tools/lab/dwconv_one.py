import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
dev = torch.device("cuda:0"); lib = L.load()
B, fthw, Cc, HD, st = 4, (16, 64, 64), 192, 96, (2, 1, 1)
cthw = [(f - 1) // s + 1 for f, s in zip(fthw, st)]
Nf = fthw[0] * fthw[1] * fthw[2]; Nc = cthw[0] * cthw[1] * cthw[2]
fine = torch.randn(B, Nf, Cc, device=dev, dtype=torch.bfloat16)
coarse = torch.randn(B, Nc, 3 * Cc, device=dev, dtype=torch.bfloat16)
w = torch.randn(HD, 27, device=dev)
g = L.DwconvGeom(); g.B, g.C, g.HD = B, Cc, HD
g.Tf, g.Hf, g.Wf = fthw; g.Tc, g.Hc, g.Wc = cthw; g.st, g.sh, g.sw = st
g.fine_batch_stride, g.fine_token_stride = Nf * Cc, Cc
g.coarse_batch_stride, g.coarse_token_stride = Nc * 3 * Cc, 3 * Cc
s = torch.cuda.current_stream().cuda_stream
dw = torch.empty(HD * 27, device=dev)
ws = torch.empty(max(16, lib.csts_dwconv_wgrad_workspace(C.byref(g))), dtype=torch.uint8, device=dev)
for _ in range(3):
    lib.csts_dwconv_wgrad(C.byref(g), fine.data_ptr(), 1, coarse.data_ptr(), 1, dw.data_ptr(), ws.data_ptr(), ws.numel(), s)
torch.cuda.synchronize()
