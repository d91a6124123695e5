"""Diagnostic: per-parameter gradient comparison of the HIP model (fp32 mode) against the reference golden."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd.config import load_yaml
from csts_amd.build import build_model
from csts_amd import train as T
from oracle import csts_oracle as O

dev = torch.device("cuda:0")
cfg = load_yaml("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "CSTS_AMD.COMPUTE", sys.argv[1] if len(sys.argv) > 1 else "fp32"])
m = build_model(cfg); m.load_state_dict(O.seeded_params(8, 256)); m.eval()
g = np.load("tests/golden/model_T8_B2.npz")
b = {k: v.to(dev) for k, v in O.synthetic_batch(2, 8, 256, seed=1000).items()}
loss, kld, nce, preds = T.compute_loss(cfg, m, b["video"], b["audio"], b["labels_hm"])
loss.backward()
named = dict(m.named_parameters())
for n, rn in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
    gr = named[n].grad
    sl = gr.flatten()[:64].cpu().double()
    ref = torch.from_numpy(g[n.replace(".", "_") + "_g"]).double()
    print(f"{n:45s} norm {float(gr.norm()):.6e} ref {rn:.6e} relnorm {abs(float(gr.norm())-rn)/max(rn,1e-12):.2e} slice_rel {float((sl-ref).norm()/(ref.norm()+1e-30)):.2e}")
tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
print("total", tot, float(g["grad_total_norm"]))
