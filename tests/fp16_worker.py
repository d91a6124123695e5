"""Child process of tests/test_gpu_fp16.py: the fp16 compute mode (CSTS_AMD.COMPUTE fp16 = libcsts_hip_f16.so + dynamic loss
scaling) on the reference fixtures.  One process runs one 16-bit type, so the fp16 checks cannot share pytest's process with the
bf16 ones.  Writes a JSON document of measured quantities to argv[1]."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csts_amd.config import load_yaml           # noqa: E402
from csts_amd.build import build_model          # noqa: E402
from csts_amd import lib as L, train as T       # noqa: E402
from oracle import csts_oracle as O             # noqa: E402  (test infrastructure)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DEV = torch.device("cuda:0")


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / b.norm())


def model_for(yaml, T_, extra=()):
    cfg = load_yaml(os.path.join(ROOT, yaml), ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", T_,
                                                "CSTS_AMD.COMPUTE", "fp16"] + list(extra))
    m = build_model(cfg)
    m.load_state_dict(O.seeded_params(T_, 256), strict=True)
    m.eval()
    return m, cfg


def train_case(yaml, T_, B, seed, fixture):
    a = np.load(os.path.join(GOLDEN, fixture), allow_pickle=False)
    m, cfg = model_for(yaml, T_)
    opt = T.construct_optimizer(m, cfg)
    assert opt.loss_scale is not None and float(opt.loss_scale) == 65536.0
    batch = {k: v.to(DEV) for k, v in O.synthetic_batch(B, T_, 256, seed=seed).items()}
    scale = float(opt.loss_scale)
    tries = 0
    while True:         # GradScaler semantics by hand (the optimizer kernels do this on the device, tested below): halve on overflow
        for p in m.parameters():
            p.grad = None
        loss, kld, nce, preds = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
        (loss * scale).backward()
        torch.cuda.synchronize()
        finite = all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
        if finite or tries > 6:
            break
        scale *= 0.5
        tries += 1
    with torch.no_grad():
        logits = m([batch["video"]], batch["audio"])
    named = dict(m.named_parameters())
    names = [str(n) for n in a["grad_names"]]
    keep = [i for i, n in enumerate(names) if n != "classifier.bias" and a["grad_norm_ref"][i] > 0]
    ours = [abs(float(named[names[i]].grad.double().norm()) / scale / a["grad_norm_ref"][i] - 1.0) for i in keep]
    theirs = np.abs(a["grad_norm_amp"][keep] / a["grad_norm_ref"][keep] - 1.0)
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))) / scale
    am = preds.reshape(B, T_, -1).argmax(-1).cpu().numpy()
    out = dict(heat=rel(preds, a["heat_ref"]), heat_ref_amp=float(a["heat_rel_l2"]), logits=rel(logits, a["logits_ref"]),
               logits_ref_amp=float(a["logits_rel_l2"]), argmax=float((am == a["argmax_ref"]).mean()), argmax_ref_amp=float(a["argmax_agree"]),
               loss_err=abs(float(loss) - float(a["loss_ref"])) / float(a["loss_ref"]),
               loss_err_ref_amp=abs(float(a["loss_amp"]) - float(a["loss_ref"])) / float(a["loss_ref"]),
               gn_median=float(np.median(ours)), gn_max=float(np.max(ours)), gn_median_ref_amp=float(np.median(theirs)),
               gn_max_ref_amp=float(theirs.max()), total_err=abs(total / float(a["grad_total_norm_ref"]) - 1.0),
               total_err_ref_amp=abs(float(a["grad_total_norm_amp"]) / float(a["grad_total_norm_ref"]) - 1.0),
               loss_scale_used=scale, ref_loss_scale=float(a["loss_scale"]), finite=bool(finite),
               half_dtype=str(L.half_dtype()), preds_dtype=str(preds.dtype))
    del m, opt
    torch.cuda.empty_cache()
    return out


def scaler_case():
    """The device-side GradScaler: a real train step (HIP graph) keeps finite weights; an injected inf gradient skips the step
    (no parameter / moment / step-count change) and halves the scale; growth_interval good steps double it."""
    m, cfg = model_for("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", 8, ["TRAIN.BATCH_SIZE", 2])
    m.train()
    opt = T.construct_optimizer(m, cfg)
    opt.scaler_cfg = (2.0, 0.5, 3)                   # growth after 3 good steps
    batch = T.synthetic_batch(2, 8, 256, 1000, DEV)
    step = T.GraphedTrainStep(cfg, m, opt, batch)
    scales, steps, losses = [], [], []
    for i in range(8):
        loss, kld, nce = step.run(batch, 1e-4)
        torch.cuda.synchronize()
        scales.append(float(opt.loss_scale)); steps.append(opt.step_count()); losses.append(float(loss))
    w0 = m.blocks[3].mlp.fc1.weight.detach().clone()
    m0 = opt.exp_avg.clone()
    # eager step with a poisoned gradient
    opt.zero_grad()
    loss, _, _, _ = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
    T.backward_scaled(loss, opt)
    m.blocks[3].mlp.fc1.weight.grad[0, 0] = float("inf")
    s_before, n_before = float(opt.loss_scale), opt.step_count()
    opt.step()
    torch.cuda.synchronize()
    out = dict(scales=scales, steps=steps, losses=losses, skipped_flag=float(opt.state_t[3]), scale_before=s_before,
               scale_after=float(opt.loss_scale), steps_before=n_before, steps_after=opt.step_count(),
               weights_unchanged=bool(torch.equal(w0, m.blocks[3].mlp.fc1.weight)), moments_unchanged=bool(torch.equal(m0, opt.exp_avg)),
               weights_finite=bool(all(torch.isfinite(p).all() for p in m.parameters())),
               scaler_state=T.scaler_of(opt).state_dict())
    return out


if __name__ == "__main__":
    res = {"T8": train_case("configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", 8, 2, 1000, "autocast_fp16_T8_B2.npz"),
           "T32_aria": train_case("configs/Aria/CSTS_Aria_Gaze_Forecast.yaml", 32, 1, 1003, "autocast_fp16_T32_B1_aria.npz"),
           "scaler": scaler_case()}
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(res, indent=1))
