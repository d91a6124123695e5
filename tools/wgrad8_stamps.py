"""Shader-clock stamps of wgrad8_kernel (diagnostics build: `make -C csts_amd/csrc stamps` -> tools/diag/libcsts_hip_stamps.so).
Workgroup 0, per 64-token k-tile: wait for this wave's LDS-DMA pieces | barrier | issue of the next refill | fragment reads + 36
MFMAs (1152 ticks of matrix-pipe issue per wave, two waves per SIMD).  The layers of tools/wgrad8_bench.py."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libcsts_hip_stamps.so")
from csts_amd import ops
dev = torch.device("cuda:0")
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 11
layers = [(1152, 384), (384, 384), (1536, 384), (384, 1536)] * nblk
prob = [(torch.randn(tokens, N, device=dev).bfloat16(), torch.randn(tokens, K, device=dev).bfloat16(), N, K) for N, K in layers]
ops.WGRAD8 = True
for it in range(3):
    for dY, X, N, K in prob:
        ops._wgq.append((dY, X, torch.empty(N, K, device=dev), torch.empty(N, device=dev), tokens, N, K))
    ops.flush_wgrads(); ops.flush_deferred()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 4096)()
raw = C.CDLL(L.LIB_PATH)
raw.csts_debug_wgrad8_stamps.argtypes = [C.c_void_p]
assert raw.csts_debug_wgrad8_stamps(buf) == 0
n = buf[0]
t = [buf[i] for i in range(1, n)]
rows = []
i = 1
prev = t[0]
while i + 4 <= len(t):
    g = t[i:i + 4]
    rows.append((g[0] - prev, g[1] - g[0], g[2] - g[1], g[3] - g[2], g[3] - prev))
    prev = g[3]
    i += 4
import statistics as st
print(f"{len(rows)} k-tiles; k-loop {t[-1] - t[0]} ticks")
for r in rows[:6]:
    print("  wait %5d | barrier %5d | issue %4d | reads + MFMAs %5d | k-tile %5d" % r)
print("median: wait %d | barrier %d | issue %d | reads + MFMAs %d | k-tile %d" % tuple(int(st.median(r[j] for r in rows[2:])) for j in range(5)))
