"""Which torch-native (aten) ops run inside one train step, and from where: counts and device time per (op, csts_amd call
site), bench configuration (b=4, 16 x 256^2, bf16), grouped weight gradients on as in the captured step."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T, ops
dev = torch.device("cuda:0")
cfg = load_yaml("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 4, "MODEL.LOSS_FUNC", "kldiv+egonce",
                                                                  "MODEL.LOSS_ALPHA", 0.05, "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16"])
torch.manual_seed(1)
m = build_model(cfg); m.train()
opt = T.construct_optimizer(m, cfg, capturable=True)
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
ops.GROUP_WGRADS = "always"
for _ in range(2):
    T.train_step(cfg, m, batch, opt, 1e-4)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    T.train_step(cfg, m, batch, opt, 1e-4)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time_total <= 0:
        continue
    if e.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
        continue                                   # count leaves only
    st = [s for s in (e.stack or []) if "csts_amd" in s]
    site = (st[0].split("/")[-1] if st else "(autograd engine)") + "  " + str([tuple(x) for x in (e.input_shapes or []) if x][:2])
    a = agg[(e.name, site)]
    a[0] += 1; a[1] += e.device_time_total
tot = sum(v[1] for v in agg.values())
print(f"torch-native device time in one step: {tot / 1e3:.3f} ms over {sum(v[0] for v in agg.values())} leaf ops")
for (name, site), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:8.3f} ms {n:5d} x  {name:28s} {site}")
