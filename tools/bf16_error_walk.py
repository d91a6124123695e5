#!/usr/bin/env python3
"""Where the bf16 compute mode's forward error comes from: the same seeded weights and batch through
the fp32 mode (== the reference to 4e-7) and the bf16 mode, relative L2 of every block output, heat maps and arg-max.

    python tools/bf16_error_walk.py [--frames 8] [--batch 2] [--seed 1000]

Environment switches of the library under test apply (e.g. CSTS_PATCH_EMBED_SPLIT=0 to see the effect of the hi+lo operand
split of the patch embeddings)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from csts_amd.build import build_model          # noqa: E402
from csts_amd.config import load_yaml           # noqa: E402
from csts_amd.model import Block                # noqa: E402
from csts_amd import ops                        # noqa: E402
from oracle import csts_oracle as O             # noqa: E402  (diagnostics tool, not the product path)

YAML = os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml")


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def run(compute, T, batch, opts=()):
    cfg = load_yaml(YAML, ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", T, "CSTS_AMD.COMPUTE", compute] + list(opts))
    m = build_model(cfg)
    m.load_state_dict(O.seeded_params(T, 256), strict=True)
    m.eval()
    taps = {}
    hooks = []
    for name, mod in m.named_modules():
        if isinstance(mod, Block):
            hooks.append(mod.register_forward_hook(lambda mod_, i, o, name=name: taps.__setitem__(name, o[0].detach().float().clone())))
    orig = ops.patch_embed
    n = [0]

    def pe(*a, **k):
        y = orig(*a, **k)
        taps["patch_embed" + ("_audio" if n[0] else "")] = y.detach().float().clone()
        n[0] += 1
        return y
    ops.patch_embed = pe
    try:
        with torch.no_grad():
            logits, v, a = m([batch["video"]], batch["audio"], return_embed=True)
    finally:
        ops.patch_embed = orig
        for h in hooks:
            h.remove()
    taps["logits"] = logits.float()
    taps["heat"] = ops.frame_softmax(logits, 2.0).float()
    taps["v_emb"], taps["a_emb"] = v.float(), a.float()
    del m
    torch.cuda.empty_cache()
    return taps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    batch = {k: v.to(dev) for k, v in O.synthetic_batch(args.batch, args.frames, 256, seed=args.seed).items()}
    ref = run("fp32", args.frames, batch)
    got = run("bf16", args.frames, batch)
    print(f"# bf16 mode vs fp32 mode, T={args.frames} B={args.batch} seed={args.seed}: relative L2 per stage output")
    for k in ref:
        print(f"{k:28s} {rel(got[k], ref[k]):.3e}")
    B, T = ref["heat"].shape[0], ref["heat"].shape[2]
    am_r = ref["heat"].reshape(B, T, -1).argmax(-1)
    am_g = got["heat"].reshape(B, T, -1).argmax(-1)
    print("argmax agreement", float((am_r == am_g).float().mean()), f"({int((am_r == am_g).sum())}/{am_r.numel()})")


if __name__ == "__main__":
    main()
