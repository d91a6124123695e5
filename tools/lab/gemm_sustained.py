"""Does a GEMM slow down under SUSTAINED load (clock) or under COLD operands?  One shape, library default kernel:
(a) burst: 8 back-to-back calls after an idle gap; (b) sustained: 3000 back-to-back calls, mean of the first / last 200;
(c) sustained with operands rotating over 24 buffer sets (~1.5 GB: nothing stays in L2 / Infinity Cache);
(d) the same GEMM interleaved with a streaming kernel of the train step (LayerNorm-sized copy) as in the real step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 1536, 384)))
algo = int(sys.argv[4]) if len(sys.argv) > 4 else 0
NS = 24
sets = [(torch.randn(M, K, device=dev).bfloat16(), (0.1 * torch.randn(N, K, device=dev)).bfloat16(),
         torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(NS)]
bias = torch.randn(N, device=dev)
def run(i): 
    A, B, C_ = sets[i]
    ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, C_, N, M, N, K, compute=L.BF16, bias=bias, algo=algo)
def timed(n, idx):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        run(idx(i)); evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
for _ in range(3): run(0)
torch.cuda.synchronize(); time.sleep(0.5)
b = timed(8, lambda i: 0)
print(f"{M}x{N}x{K} algo {algo}: burst (same operands)       : " + " ".join(f"{x:.1f}" for x in b))
time.sleep(0.5)
s = timed(3000, lambda i: 0)
print(f"sustained same operands: first 200 {sum(s[:200])/200:.1f} us, last 200 {sum(s[-200:])/200:.1f} us")
time.sleep(0.5)
s = timed(3000, lambda i: i % NS)
print(f"sustained rotating {NS} sets: first 200 {sum(s[:200])/200:.1f} us, last 200 {sum(s[-200:])/200:.1f} us")
x = torch.randn(64 << 20, device=dev)
y = torch.empty_like(x)
time.sleep(0.5)
evs = []
for i in range(600):
    y.copy_(x)                       # 512 MB of streaming traffic between two GEMMs: caches hold nothing of the operands
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(i % NS); e1.record(); evs.append((e0, e1))
torch.cuda.synchronize()
t = [a.elapsed_time(b_) * 1e3 for a, b_ in evs]
print(f"interleaved with 512 MB streaming copies: first 100 {sum(t[:100])/100:.1f} us, last 100 {sum(t[-100:])/100:.1f} us")
