"""What a small kernel pays for the state of its operands: csts_layernorm_fwd on 8192 x 384 (fp32 in, bf16 out) inside captured graphs --
(a) back to back on the same buffers (hot), (b) each launch after a kernel that has just WRITTEN its input (producer -> consumer, as in the
step), (c) each launch after 600 MB of unrelated traffic (the memory-side cache holds 256 MB), (d) = (b) + (c): input written, then
unrelated traffic, then the launch.  Times are per LayerNorm launch: the chain with it minus the same chain without it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import lib as L
dev = torch.device("cuda:0"); lib = L.load()
rows, Cc, N = 8192, 384, 40
x = torch.randn(rows, Cc, device=dev); src = torch.randn(rows, Cc, device=dev)
g = torch.randn(Cc, device=dev); b = torch.randn(Cc, device=dev)
y = torch.empty(rows, Cc, device=dev, dtype=torch.bfloat16); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
big_a = torch.randn(75_000_000, device=dev); big_b = torch.empty_like(big_a)          # 300 MB read + 300 MB written
def ln():
    lib.csts_layernorm_fwd(x.data_ptr(), 0, g.data_ptr(), b.data_ptr(), y.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(), rows, Cc, 1e-6, torch.cuda.current_stream().cuda_stream)
def produce(): x.copy_(src)
def unrelated(): big_b.copy_(big_a)
def chain(fs):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in fs: f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(N):
                for f in fs: f()
        for _ in range(2): gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 5 / N
for name, pre in (("(a) hot, back to back", []), ("(b) input just written by the previous kernel", [produce]),
                  ("(c) after 600 MB of unrelated traffic", [unrelated]), ("(d) input written, 600 MB of unrelated traffic, then the launch", [produce, unrelated])):
    with_ln, without = chain(pre + [ln]), (chain(pre) if pre else 0.0)
    print(f"{name:68s} {with_ln - without:6.1f} us per LayerNorm launch   (chain {with_ln:.1f} us, without it {without:.1f} us)")
