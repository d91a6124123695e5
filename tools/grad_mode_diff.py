#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient of the bf16 compute mode against the fp32 (parity) mode of the SAME HIP model, same
weights, same batch -- cosine and norm ratio, in module order.  Locates the block where a bf16-mode backward kernel goes
wrong (the fp32 mode is pinned to the reference by tests/test_gpu_model.py)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csts_amd.config import load_yaml          # noqa: E402
from csts_amd.build import build_model         # noqa: E402
from csts_amd import train as T                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--thr", type=float, default=0.98)
    ap.add_argument("--one-stream", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    yaml = os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml")
    grads, state = {}, None
    batch = T.synthetic_batch(a.batch, a.frames, 256, 1000, dev)
    for mode in ("fp32", "bf16"):
        cfg = load_yaml(yaml, ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", a.frames, "CSTS_AMD.COMPUTE", mode])
        torch.manual_seed(0)
        m = build_model(cfg)
        if state is None:
            # non-trivial LayerNorm affines / biases so that every gradient path carries signal
            g = torch.Generator(device=dev).manual_seed(5)
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if p.dim() == 1:
                        p.add_(0.1 * torch.randn(p.shape, generator=g, device=dev))
            state = {k: v.clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(state)
        m.eval()
        if a.one_stream:
            m.two_streams = False
        loss, kld, nce, _ = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
        loss.backward()
        torch.cuda.synchronize()
        print(f"{mode}: loss {float(loss):.6f} kld {float(kld):.6f} nce {float(nce):.6f}")
        grads[mode] = {n: p.grad.double().flatten().cpu() for n, p in m.named_parameters()}
        del m
        torch.cuda.empty_cache()
    nbad = 0
    for n, g32 in grads["fp32"].items():
        g16 = grads["bf16"][n]
        cos = float((g32 * g16).sum() / (g32.norm() * g16.norm()).clamp_min(1e-300))
        ratio = float(g16.norm() / g32.norm().clamp_min(1e-300))
        flag = "" if (cos >= a.thr and 0.9 < ratio < 1.1) else "   <<<<"
        nbad += bool(flag)
        print(f"{n:48s} cos {cos:8.5f}  |bf16|/|fp32| {ratio:8.4f}  |fp32| {float(g32.norm()):.3e}{flag}")
    print(f"{nbad} of {len(grads['fp32'])} parameters flagged")


if __name__ == "__main__":
    main()
