"""Where do the small device copies / fills / adds of one train step come from?  Eager step under torch.profiler with python
stacks: aten ops that launch a kernel, grouped by the innermost csts_amd source line."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T, ops
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"),
                ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "MODEL.LOSS_ALPHA", 0.05, "DATA.NUM_FRAMES", 16, "CSTS_AMD.COMPUTE", "bf16", "TRAIN.BATCH_SIZE", 4])
torch.manual_seed(1)
dev = torch.device("cuda:0")
model = build_model(cfg); model.train()
batch = T.synthetic_batch(4, 16, 256, 1000, dev)
opt = T.construct_optimizer(model, cfg, capturable=True)
ops.GROUP_WGRADS = "always"
for _ in range(2):
    T.train_step(cfg, model, batch, opt, 1e-4)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    T.train_step(cfg, model, batch, opt, 1e-4)
    torch.cuda.synchronize()
agg = collections.Counter(); dur = collections.Counter()
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::_foreach_copy_", "aten::clone", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::zeros", "aten::empty_like")
for ev in prof.events():
    if ev.name in WATCH and ev.device_time_total > 0 or ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::add", "aten::add_", "aten::mul"):
        site = "?"
        for fr in ev.stack:
            if "csts_amd" in fr and "site-packages" not in fr:
                site = fr.strip()
                break
        agg[(ev.name, site)] += 1
        dur[(ev.name, site)] += ev.device_time_total
for (name, site), n in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:45]:
    print(f"{n:4d} x {name:22s} {dur[(name, site)]:9.1f} us  {site[-110:]}")
