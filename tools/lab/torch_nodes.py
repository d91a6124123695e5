"""Which lines of the package still launch torch-native kernels inside the train step?  One eager step (after warm-up) under torch.profiler
with Python stacks: every aten op that launches a device kernel, with the innermost csts_amd frame.  usage: python tools/lab/torch_nodes.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T, ops
dev = torch.device("cuda", 0)
cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"), ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 4, "MODEL.LOSS_FUNC", "kldiv+egonce", "MODEL.LOSS_ALPHA", 0.05,
                "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16"])
torch.manual_seed(0)
model = build_model(cfg)
model.train(True)
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
opt = T.construct_optimizer(model, cfg, capturable=True)
ops.GROUP_WGRADS = "always"
for _ in range(3):
    T.train_step(cfg, model, batch, opt, 1e-4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    T.train_step(cfg, model, batch, opt, 1e-4)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    kids = [k for k in getattr(ev, "kernels", [])]
    if not kids:
        continue
    if any(c.name.startswith("aten::") and getattr(c, "kernels", []) for c in ev.cpu_children):
        continue                       # count the innermost op only
    frame = next((f for f in ev.stack if "csts_amd" in f), ev.stack[0] if ev.stack else "?")
    cnt[(ev.name, ",".join(sorted({k.name[:60] for k in kids})), frame[-110:])] += 1
for (name, ker, frame), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name:28s} {ker:62s} {frame}")
