"""Micro-benchmark of the residual-path resampling kernels on the CSTS shapes (b = 4, 16 x 256^2), straight through the C ABI:
max-pool skip forward / backward (fp32 stream) and trilinear up-sampling backward."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
dev = torch.device("cuda:0")
lib = L.load()
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def geom(B, Cc, thw_in, thw_out, stride):
    g = L.PoolGeom()
    g.B, g.C = B, Cc
    g.Ti, g.Hi, g.Wi = thw_in
    g.To, g.Ho, g.Wo = thw_out
    g.st, g.sh, g.sw = stride
    return g
B = 4
s = torch.cuda.current_stream().cuda_stream
print("maxpool skip (fwd us, bwd us):")
for name, thw, Cc in [("b1", (8, 64, 64), 192), ("b3", (8, 32, 32), 384), ("b14", (8, 16, 16), 768)]:
    out = [thw[0], thw[1] // 2, thw[2] // 2]
    g = geom(B, Cc, thw, out, (1, 2, 2))
    x = torch.randn(B, thw[0] * thw[1] * thw[2], Cc, device=dev)
    y = torch.empty(B, out[0] * out[1] * out[2], Cc, device=dev); arg = torch.empty(y.shape, dtype=torch.uint8, device=dev)
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    tf = timeit(lambda: lib.csts_maxpool_fwd(C.byref(g), x.data_ptr(), 0, y.data_ptr(), arg.data_ptr(), s))
    tb = timeit(lambda: lib.csts_maxpool_bwd(C.byref(g), dy.data_ptr(), 0, arg.data_ptr(), dx.data_ptr(), s))
    print(f"  {name:4s} {tf:8.1f} {tb:8.1f}")
print("trilinear backward (us):")
for name, thw, st, Cc in [("dec1", (8, 8, 8), (1, 2, 2), 768), ("dec2", (8, 16, 16), (1, 2, 2), 768), ("dec3", (8, 32, 32), (1, 2, 2), 384),
                          ("dec4", (8, 64, 64), (2, 1, 1), 192), ("head", (8, 64, 64), (2, 1, 1), 96)]:
    out = [t * k for t, k in zip(thw, st)]
    g = geom(B, Cc, thw, out, st)
    dy = torch.randn(B, out[0] * out[1] * out[2], Cc, device=dev); dx = torch.empty(B, thw[0] * thw[1] * thw[2], Cc, device=dev)
    tb = timeit(lambda: lib.csts_trilinear_bwd(C.byref(g), dy.data_ptr(), 0, dx.data_ptr(), 0, s))
    print(f"  {name:4s} {tb:8.1f}")
