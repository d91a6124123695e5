"""Why the step's GEMMs run 1.5x slower than the same shapes timed back to back: cold operands.  Times one NT GEMM
(HIP events around the single launch) after (a) nothing (hot: same buffers as the previous call), (b) a 1 GiB memset that
evicts L2 and the Infinity Cache, (c) the memset followed by a read of the weights only, (d) the memset followed by a read
of the activations only.  usage: gemm_cold_lab.py [M N K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (8192, 1536, 384)
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.zeros(1, device=dev)

def run(): ops.gemm(L.GEMM_NT, A, 0, K, B, 0, K, Cm, N, M, N, K, compute=L.BF16)
def timed(prep):
    ts = []
    for _ in range(12):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200_000)          # let the preparation drain; the launch is queued behind the spin
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
evict = lambda: big.fill_(1)
def evict_then_touch(t):
    def f():
        big.fill_(1)
        sink.add_(t.float().sum() * 0)      # reads t through L2 / Infinity Cache
    return f
for _ in range(3): run()
print(f"NT {M}x{N}x{K}  (us, median of 12)")
print(f"  hot (back to back)            {timed(lambda: None):7.1f}")
print(f"  after 1 GiB eviction          {timed(evict):7.1f}")
print(f"  evicted, weights re-read      {timed(evict_then_touch(B)):7.1f}")
print(f"  evicted, activations re-read  {timed(evict_then_touch(A)):7.1f}")
print(f"  evicted, output re-read       {timed(evict_then_touch(Cm)):7.1f}")
