"""Gradients of one train step with and without the pre-scaled bf16 gradient copies (ops.SCALED_GRAD_COPY): must be bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T, ops
dev = torch.device("cuda:0")
cfg = load_yaml("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "TRAIN.BATCH_SIZE", 2, "MODEL.LOSS_FUNC", "kldiv+egonce",
                                                                  "MODEL.LOSS_ALPHA", 0.05, "DATA.NUM_FRAMES", 8, "CSTS_AMD.COMPUTE", "bf16"])
torch.manual_seed(3)
m = build_model(cfg).train()
batch = T.synthetic_batch(2, 8, 256, 11, dev)
km = None
ops.GROUP_WGRADS = os.environ.get("GW", "always")
res = {}
for mode in (False, True):
    ops.SCALED_GRAD_COPY = mode
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(5)                      # same drop-path draws
    loss, *_ = T.train_step(cfg, m, batch)
    torch.cuda.synchronize()
    res[mode] = (float(loss), {n: p.grad.clone() for n, p in m.named_parameters()})
print("loss", res[False][0], res[True][0])
bad = [(n, (res[False][1][n] - res[True][1][n]).abs().max().item()) for n in res[False][1] if not torch.equal(res[False][1][n], res[True][1][n])]
print(len(bad), "tensors differ of", len(res[False][1]))
for n, d in bad[:20]:
    print(" ", n, d)
