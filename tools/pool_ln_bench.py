"""csts_pool_ln_fwd (k | v conv-pool + LayerNorm(hd)) on the CSTS shapes (b = 4, 16 x 256^2, bf16), through the C ABI."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
dev = torch.device("cuda:0"); lib = L.load(); s = torch.cuda.current_stream().cuda_stream
B = 4
SH = [("b0.kv", (8, 64, 64), 96, 96, (1, 8, 8), 2), ("b1.kv", (8, 64, 64), 192, 96, (1, 4, 4), 2), ("b1.q", (8, 64, 64), 192, 96, (1, 2, 2), 1),
      ("b3.kv", (8, 32, 32), 384, 96, (1, 2, 2), 2), ("b4-13.kv", (8, 16, 16), 384, 96, (1, 2, 2), 2), ("b14.kv", (8, 16, 16), 768, 96, (1, 1, 1), 2),
      ("b15.kv", (8, 8, 8), 768, 96, (1, 1, 1), 2)]
for name, fthw, Cc, HD, st, ns in SH:
    Nf = fthw[0] * fthw[1] * fthw[2]
    cthw = [(f - 1) // k + 1 for f, k in zip(fthw, st)]
    Nc = cthw[0] * cthw[1] * cthw[2]
    H = Cc // HD
    qkv = torch.randn(B, Nf, 3 * Cc, device=dev, dtype=torch.bfloat16)
    w = [torch.randn(HD, 27, device=dev) for _ in range(ns)]
    gam = [torch.randn(HD, device=dev) for _ in range(ns)]; bet = [torch.randn(HD, device=dev) for _ in range(ns)]
    c = torch.empty(ns, B, Nc, Cc, device=dev, dtype=torch.bfloat16); y = torch.empty_like(c)
    mean = torch.empty(ns, B * Nc * H, device=dev); rstd = torch.empty_like(mean)
    pa = L.PoolLnArgs()
    g = pa.geom
    g.B, g.C, g.HD = B, Cc, HD
    g.Tf, g.Hf, g.Wf = fthw; g.Tc, g.Hc, g.Wc = cthw; g.st, g.sh, g.sw = st
    g.fine_batch_stride, g.fine_token_stride = Nf * 3 * Cc, 3 * Cc
    g.coarse_batch_stride, g.coarse_token_stride = Nc * Cc, Cc
    pa.nslots, pa.dt, pa.eps = ns, 1, 1e-5
    for i in range(ns):
        pa.fine[i] = qkv.data_ptr() + (i + 1) * Cc * 2
        pa.weight[i], pa.gamma[i], pa.beta[i] = w[i].data_ptr(), gam[i].data_ptr(), bet[i].data_ptr()
        pa.conv_out[i], pa.y[i], pa.mean[i], pa.rstd[i] = c[i].data_ptr(), y[i].data_ptr(), mean[i].data_ptr(), rstd[i].data_ptr()
    f = lambda: lib.csts_pool_ln_fwd(C.byref(pa), s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us   checksum {float(y.float().abs().sum()):.5e}")
