"""LayerNorm forward / backward on the residual-stream shapes of one CSTS step (b = 4, 16 x 256^2, bf16 mode: fp32 stream in, bf16
normalised rows out; backward: bf16 dy, fp32 x / residual-branch gradient in, fp32 dx + bf16 copy out), through the C ABI."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
dev = torch.device("cuda:0"); lib = L.load(); s = torch.cuda.current_stream().cuda_stream
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("rows C | fwd us GB/s | bwd us GB/s")
for rows, Cc in [(131072, 96), (262144, 96), (131072, 192), (32768, 192), (262144, 192), (32768, 384), (8192, 384), (131072, 384), (8192, 768), (2048, 768), (32768, 768)]:
    x = torch.randn(rows, Cc, device=dev); g = torch.randn(Cc, device=dev); b = torch.randn(Cc, device=dev)
    y = torch.empty(rows, Cc, device=dev, dtype=torch.bfloat16); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    tf = timeit(lambda: lib.csts_layernorm_fwd(x.data_ptr(), 0, g.data_ptr(), b.data_ptr(), y.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(), rows, Cc, 1e-6, s))
    dy = torch.randn(rows, Cc, device=dev, dtype=torch.bfloat16); add = torch.randn(rows, Cc, device=dev)
    dx = torch.empty_like(x); dx16 = torch.empty_like(y)
    nb = lib.csts_layernorm_bwd_workspace(rows, Cc)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    tb = timeit(lambda: lib.csts_layernorm_bwd_ex(dy.data_ptr(), None, 1, x.data_ptr(), 0, g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), 0,
                                                  add.data_ptr(), dx16.data_ptr(), None, 1, None, None, ws.data_ptr(), nb, rows, Cc, s))
    bf, bb = rows * Cc * 6, rows * Cc * 16
    print(f"{rows:7d} {Cc:4d} | {tf:7.1f} {bf / tf / 1e3:6.0f} | {tb:7.1f} {bb / tb / 1e3:6.0f}")
