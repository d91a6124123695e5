"""The real grouped weight-gradient kernels in isolation: the Linear layers of the 384-channel stage (11 blocks x
{qkv 1152x384, proj 384x384, fc1 1536x384, fc2 384x1536}, 8192 tokens) through csts_amd.ops.flush_wgrads, 192 x 384 tiles
(CSTS_WGRAD8=1) against 128 x 128 / 256 x 128 tiles (=0): TF/s of one grouped launch set, correctness of one layer."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 11
layers = [(1152, 384), (384, 384), (1536, 384), (384, 1536)] * nblk
prob = []
for (N, K) in layers:
    dY = torch.randn(tokens, N, device=dev).bfloat16()
    X = torch.randn(tokens, K, device=dev).bfloat16()
    prob.append((dY, X, N, K))
flop = sum(2.0 * tokens * N * K for _, _, N, K in prob)
for w8 in (True, False, True, False):
    ops.WGRAD8 = w8
    best = 1e9
    for it in range(4):
        outs = []
        for dY, X, N, K in prob:
            dW = torch.empty(N, K, device=dev)
            db = torch.empty(N, device=dev)
            ops._wgq.append((dY, X, dW, db, tokens, N, K))
            outs.append((dW, db))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ops.flush_wgrads()
        ops.flush_deferred()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    dY, X, N, K = prob[2]
    ref = dY.float().t() @ X.float()
    err = ((outs[2][0] - ref).norm() / ref.norm()).item()
    errb = ((outs[2][1] - dY.float().sum(0)).norm() / dY.float().sum(0).norm()).item()
    print(f"WGRAD8={int(w8)}: {best * 1e3:8.1f} us for {flop / 1e9:.0f} GFLOP = {flop / best / 1e9:7.1f} TF/s   (fc1 dW rel err {err:.1e}, db {errb:.1e})")
