"""Shader clock inside the replayed train step: probes (tools/clock_probe.hip) are launched after chosen blocks of the forward and, through
autograd hooks, the backward pass; the captured graph is replayed and the probes' clocks printed, next to the same probe on an idle GPU."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(ROOT, "tools/diag/libclock_probe.so"))
lib.clock_probe.argtypes = [C.c_void_p, C.c_void_p]
buf = torch.zeros(256, device=dev)
names = {}
def probe(tag):
    i = names.setdefault(tag, len(names))            # a fixed slot per tag: warm-up, capture pass and replays write the same word
    lib.clock_probe(buf.data_ptr() + 4 * i, torch.cuda.current_stream().cuda_stream)
idle = torch.zeros(8, device=dev)
for i in range(8):
    lib.clock_probe(idle.data_ptr() + 4 * i, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("idle GPU, eight probes back to back (MHz):", [round(v) for v in idle.tolist()])
cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"),
                ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 4, "MODEL.LOSS_FUNC", "kldiv+egonce", "MODEL.LOSS_ALPHA", 0.05, "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16"])
torch.manual_seed(1)
m = build_model(cfg); m.train()
m.two_streams = False                      # one stream: the probes sit between the kernels they are meant to sample
opt = T.construct_optimizer(m, cfg, capturable=True)
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        probe("fwd after " + tag)
        return x.view_as(x)
    @staticmethod
    def backward(ctx, g):
        probe("bwd before " + ctx.tag)
        return g, None
def hook(tag):
    def f(mod, inp, out):
        if isinstance(out, tuple):
            return (Probe.apply(out[0], tag),) + tuple(out[1:])
        return Probe.apply(out, tag)
    return f
for nm in ("blocks.0", "blocks.2", "blocks.5", "blocks.9", "blocks.13", "blocks.15", "decode_block2", "decode_block4"):
    mod = m
    for part in nm.split("."):
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    mod.register_forward_hook(hook(nm))
step = T.GraphedTrainStep(cfg, m, opt, batch)
for _ in range(10):
    step.run(batch, 1e-4)
torch.cuda.synchronize()
vals = buf.tolist()
for t, i in names.items():
    print(f"{t:32s} {vals[i]:7.0f} MHz")
