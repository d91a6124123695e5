import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T
dev = torch.device("cuda:0")
cfg = load_yaml("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", 8])
m = build_model(cfg); m.train()
opt = T.construct_optimizer(m, cfg)
batch = T.synthetic_batch(2, 8, 256, 1, dev)
T.train_step(cfg, m, batch, opt, 1e-4)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    T.train_step(cfg, m, batch, opt, 1e-4)
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::ones_like"):
        st = [s for s in (e.stack or []) if "csts_amd" in s or "torch/nn/utils" in s or "optim" in s]
        cnt[(e.name, st[0] if st else (e.stack[0] if e.stack else "?"))] += 1
for k, v in cnt.most_common(12):
    print(v, k)
