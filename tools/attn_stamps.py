"""Shader-clock stamps of the attention dK/dV kernel (diagnostics build: `make -C csts_amd/csrc stamps` ->
tools/diag/libcsts_hip_stamps.so).  Workgroup (0,0,0), per 64-query tile: cycles from the previous tile's end to this tile's
body (staging stores + next-tile loads + barrier), then per 32-query unit: S/dP MFMAs | softmax backward | dV/dK MFMAs.
usage: attn_stamps.py B H Nq Nk"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", os.environ.get("STAMPS_LIB", "libcsts_hip_stamps.so"))
dev = torch.device("cuda:0")
B, H, Nq, Nk = [int(v) for v in sys.argv[1:5]]
hd = 96
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
lib.csts_attn_fwd(C.byref(a), s)
for _ in range(3):
    lib.csts_attn_bwd(C.byref(a), ws.data_ptr(), ws.numel(), s)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 4096)()
raw = C.CDLL(L.LIB_PATH)
raw.csts_debug_attn_stamps.argtypes = [C.c_void_p]
assert raw.csts_debug_attn_stamps(buf) == 0
n = buf[0]
t = [buf[i] for i in range(1, n)]
print(f"{n - 1} stamps; kernel body {t[-1] - t[0]} cycles" + (f"; {buf[4095]} ticks of the 100 MHz clock -> shader clock {(t[-1] - t[0]) / buf[4095] * 0.1:.2f} GHz" if buf[4095] else ""))
# LDS-DMA tile loop (round 5, SPREAD form): start | per tile: loop top, body begins, 8 block starts (SC0 SC1 PV0 SC2 PV1 SC3 PV2 PV3),
# body done, after body | end.  The wait for the next tile + barrier sits between "after body" and the next "loop top".
per = 12
rows = []
i = 1
prev = t[0]
while i + per <= len(t) - 1:
    g = t[i:i + per]
    blocks = [g[3 + j] - g[2 + j] for j in range(7)] + [g[10] - g[9]]
    rows.append(dict(sync=g[0] - prev, body=g[10] - g[1], tile=g[11] - prev, blocks=blocks))
    prev = g[11]
    i += per
import statistics as st
keys = ["sync", "body", "tile"]
print("per 128-query tile, median shader-clock ticks: wait for the next tile + barrier | MFMA stream (96 MFMAs per wave = 3072 ticks at "
      "the matrix pipe's issue rate) | whole tile")
print("  " + " | ".join(f"{k} {int(st.median(r[k] for r in rows[1:]))}" for k in keys), f"({len(rows)} tiles)")
print("  blocks SC0 SC1 PV0 SC2 PV1 SC3 PV2 PV3 (12 MFMAs = 384 ticks each): " +
      " ".join(str(int(st.median(r["blocks"][j] for r in rows[1:]))) for j in range(8)))
