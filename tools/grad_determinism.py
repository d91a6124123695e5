"""Diagnostic: are gradients of two identical train steps bit-identical?  Which parameters differ between modes?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T, ops

dev = torch.device("cuda:0")
cfg = load_yaml("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "CSTS_AMD.COMPUTE", "bf16"])
m = build_model(cfg)
torch.manual_seed(0)          # default init of the model (no oracle here: the oracle is test infrastructure)
m.eval()
batch = T.synthetic_batch(2, 8, 256, 77, dev)


def grads():
    T.train_step(cfg, m, batch)
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}


def cmp(a, b, tag):
    worst = []
    bad = [(int((~torch.isclose(a[n], b[n], rtol=1e-4, atol=1e-7)).sum()), n) for n in a]
    bad = sorted([x for x in bad if x[0] > 0], reverse=True)
    print(tag, "params failing allclose:", len(bad), bad[:8], flush=True)
    for n in a:
        if n.endswith("norm_k.bias"):
            continue
        d = (a[n] - b[n]).abs().max().item()
        s = b[n].abs().max().item() + 1e-30
        worst.append((d / s, n))
    worst.sort(reverse=True)
    print(tag, "max rel-to-max diff:", [(f"{w:.2e}", n) for w, n in worst[:6]], flush=True)


g0 = grads(); g1 = grads()
cmp(g1, g0, "defer on, run1 vs run0:")
ops.DEFER_REDUCTIONS = False
g2 = grads(); g3 = grads()
cmp(g3, g2, "defer off, run1 vs run0:")
cmp(g2, g0, "defer off vs on:")
