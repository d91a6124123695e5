"""Patch-embed im2col on the CSTS shapes (b = 4, 16 x 256^2): video (3 channels) and audio (1 channel), fp32 in, bf16 out."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
dev = torch.device("cuda:0"); lib = L.load(); s = torch.cuda.current_stream().cuda_stream
for name, cin in (("video", 3), ("audio", 1)):
    g = L.Im2colGeom()
    g.B, g.Cin, g.T, g.H, g.W = 4, cin, 16, 256, 256
    for i, v in enumerate((3, 7, 7)): g.kernel[i] = v
    for i, v in enumerate((2, 4, 4)): g.stride[i] = v
    for i, v in enumerate((1, 3, 3)): g.padding[i] = v
    g.To, g.Ho, g.Wo = 8, 64, 64
    K = cin * 147
    g.Kpad = (K + 63) // 64 * 64
    x = torch.randn(4, cin, 16, 256, 256, device=dev)
    col = torch.empty(4 * 8 * 64 * 64, g.Kpad, device=dev, dtype=torch.bfloat16)
    f = lambda: lib.csts_im2col(C.byref(g), x.data_ptr(), 0, col.data_ptr(), 1, s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: Kpad {g.Kpad}, {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, checksum {float(col.float().abs().sum()):.6e}")
