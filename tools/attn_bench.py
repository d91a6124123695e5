"""Micro-benchmark of the fused attention kernels on the CSTS shapes (b=4, 16x256^2, bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
# name, B, H, Nq, Nk, hd
SH = [("blocks.0", 4, 1, 32768, 512, 96), ("blocks.1", 4, 2, 8192, 2048, 96), ("blocks.3", 4, 4, 2048, 2048, 96),
      ("blocks.4-12", 4, 4, 2048, 512, 96), ("blocks.14", 4, 8, 512, 2048, 96), ("dec2", 4, 4, 8192, 128, 192),
      ("dec3", 4, 4, 32768, 128, 96), ("dec4", 4, 2, 65536, 128, 96)]
import ctypes as C
def run(B, H, Nq, Nk, hd, bwd):
    Cc = H * hd
    q = torch.randn(B, Nq, Cc, device=dev, dtype=torch.bfloat16); k = torch.randn(B, Nk, Cc, device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k); o = torch.empty_like(q); lse = torch.empty(B, H, Nq, device=dev)
    do = torch.randn_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(k); delta = torch.empty_like(lse)
    a = L.AttnArgs()
    a.Q, a.K, a.V, a.O, a.LSE = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    a.dO, a.delta, a.dQ, a.dK, a.dV = do.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = L.BF16, B, H, Nq, Nk, hd
    sq = (C.c_int64 * 3)(Nq * Cc, Cc, hd); sk = (C.c_int64 * 3)(Nk * Cc, Cc, hd)
    a.q_strides = sq; a.o_strides = sq; a.do_strides = sq; a.dq_strides = sq
    a.k_strides = sk; a.v_strides = sk; a.dk_strides = sk; a.dv_strides = sk
    a.scale = hd ** -0.5
    lib = L.load(); s = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(16, lib.csts_attn_bwd_workspace(C.byref(a))), dtype=torch.uint8, device=dev)
    f = (lambda: lib.csts_attn_bwd(C.byref(a), ws.data_ptr(), ws.numel(), s)) if bwd else (lambda: lib.csts_attn_fwd(C.byref(a), s))
    lib.csts_attn_fwd(C.byref(a), s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
print(f"{'shape':12s} {'fwd us':>8s} {'TF/s':>7s} {'bwd us':>8s} {'TF/s':>7s}")
for name, B, H, Nq, Nk, hd in SH:
    if only and name != only: continue
    fl = 4.0 * B * H * Nq * Nk * hd
    tf = run(B, H, Nq, Nk, hd, False); tb = run(B, H, Nq, Nk, hd, True)
    print(f"{name:12s} {tf:8.1f} {fl/tf/1e6:7.1f} {tb:8.1f} {2.5*fl/tb/1e6:7.1f}")
