"""Shader-clock stamps of the attention dK/dV kernel (diagnostics build: `make -C csts_amd/csrc stamps` ->
tools/diag/libcsts_hip_stamps.so).  Workgroup (0,0,0), per 64-query tile: cycles from the previous tile's end to this tile's
body (staging stores + next-tile loads + barrier), then per 32-query unit: S/dP MFMAs | softmax backward | dV/dK MFMAs.
usage: attn_stamps.py B H Nq Nk"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libcsts_hip_stamps.so")
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
print(f"{n - 1} stamps; kernel body {t[-1] - t[0]} cycles")
# layout: start | per tile: body_begin, (mfma1, softmax) x QT, body_end | end
QT = 2
per = 2 + 2 * QT + 2
i = 1
tile = 0
prev_end = t[0]
rows = []
while i + per <= len(t) - 1 + 1 and tile < 10000:
    seg = t[i:i + per]
    if len(seg) < per: break
    stage = seg[0] - prev_end
    units = []
    last = seg[0]
    for u in range(QT):
        m1 = seg[1 + 2 * u] - last
        sm = seg[2 + 2 * u] - seg[1 + 2 * u]
        # dV/dK MFMAs of unit u run until the next unit's S/dP stamp; reported with it, except for the last unit
        units.append((m1, sm))
        last = seg[2 + 2 * u]
    body_end = seg[per - 3]
    tail = body_end - seg[per - 4]
    rows.append((stage, units, tail, body_end - seg[0], seg[per - 2] - body_end, seg[per - 1] - seg[per - 2]))
    prev_end = seg[per - 1]
    i += per
    tile += 1
for r in rows[:6] + rows[-2:]:
    print(f"  staging+barrier {r[0]:6d} | " + " | ".join(f"(pv of previous +) S/dP {m:5d}, softmax {sm:5d}" for m, sm in r[1]) + f" | last dV/dK {r[2]:5d} | body {r[3]} | wait for prefetch + ds_write {r[4]} | issue next loads {r[5]}")
import statistics as st
print("median: wait+ds_write", st.median(r[4] for r in rows), "issue", st.median(r[5] for r in rows), "barrier", st.median(r[0] for r in rows), "body", st.median(r[3] for r in rows),
      "unit0 S/dP", st.median(r[1][0][0] for r in rows), "softmax", st.median(r[1][0][1] for r in rows),
      "unit1 pv+S/dP", st.median(r[1][1][0] for r in rows), "softmax", st.median(r[1][1][1] for r in rows), "last pv", st.median(r[2] for r in rows))
