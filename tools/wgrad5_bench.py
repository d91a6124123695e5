"""The grouped weight gradients of the THIN layers in isolation: the problem list of one train step (written by
CSTS_WGRAD_DUMP=file python bench.py ...: dtype tokens N K lda ldb has_bias per line) through csts_amd.ops.flush_wgrads with the 96 x 96
per-wave class (CSTS_WGRAD5=1) against the 128-wide classes (=0): time of the launches, bytes that must move, correctness of two layers.
usage: wgrad5_bench.py problems.txt all|min96|wide [chunk ...]   (min96: layers with min(N, K) == 96; wide: the others)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
which = sys.argv[2] if len(sys.argv) > 2 else "all"
chunks = [int(v) for v in sys.argv[3:]] or [ops.WGRAD5_CHUNK]
prob = []
for dt, tokens, N, K, lda, ldb, hb in rows:
    tokens, N, K, lda, ldb = int(tokens), int(N), int(K), int(lda), int(ldb)
    if dt != "h16" or N % 96 or K % 96 or (N % 192 == 0 and K % 384 == 0) or tokens % 16:
        continue
    if (which == "min96" and min(N, K) != 96) or (which == "wide" and min(N, K) == 96):
        continue
    prob.append((torch.randn(tokens, N, device=dev).bfloat16(), torch.randn(tokens, K, device=dev).bfloat16(), tokens, N, K, hb == "1"))
byts = sum(t * (N + K) * 2 + N * K * 4 for _, _, t, N, K, _ in prob)
flop = sum(2.0 * t * N * K for _, _, t, N, K, _ in prob)
print(f"{len(prob)} thin problems, {byts / 1e6:.0f} MB, {flop / 1e9:.0f} GFLOP")


def enqueue():
    outs = []
    for dY, X, t, N, K, hb in prob:
        dW = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev) if hb else None
        ops._wgq.append((dY, X, dW, db, t, N, K))
        outs.append((dW, db))
    ops.flush_wgrads()
    ops.flush_deferred()
    return outs


def run(w5, chunk):
    """device time of the launches: the flush is captured into a HIP graph and replayed (an eager flush spends ~0.4 ms on the host
    building item tables, which HIP events around it would count)"""
    ops.WGRAD5, ops.WGRAD5_CHUNK = w5, chunk
    enqueue()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enqueue()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = enqueue()
    best = 1e9
    for it in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    errs = []
    for i in (0, len(prob) // 2):
        dY, X, t, N, K, hb = prob[i]
        ref = dY.float().t() @ X.float()
        errs.append(((outs[i][0] - ref).norm() / ref.norm()).item())
        if hb:
            errs.append(((outs[i][1] - dY.float().sum(0)).norm() / dY.float().sum(0).norm()).item())
    return best, errs


for rnd in range(2):
    b, e = run(False, 4096)
    print(f"128-wide classes      : {b * 1e3:8.1f} us  {byts / b / 1e9:6.2f} TB/s   errs {' '.join(f'{x:.1e}' for x in e)}")
    for c in chunks:
        b, e = run(True, c)
        print(f"96 x 96 per wave, {c:5d}: {b * 1e3:8.1f} us  {byts / b / 1e9:6.2f} TB/s   errs {' '.join(f'{x:.1e}' for x in e)}")
