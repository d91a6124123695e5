#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Build-container tooling only: needs /root/reference, never runs on the GPU box,
never imported by the product.  Run as

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py

Recipe follows SURVEY.md section 8(c) / Appendix C: throw-away stubs for the
python packages the reference imports but this image lacks (fvcore, iopath,
fairscale, ipdb, simplejson), then the reference modules are used unmodified.
Weights come from ``oracle.csts_oracle.seeded_tensor`` (a documented per-name
generator) loaded into the reference modules with ``load_state_dict(strict=True)``,
so only inputs/outputs are stored, never weights or reference source.
"""
import ast
import copy
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
sys.dont_write_bytecode = True


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install_stubs():
    for n in ("fvcore", "fvcore.common", "iopath", "iopath.common", "fairscale", "fairscale.nn", "ipdb",
              "simplejson"):
        _mod(n)

    class Registry:
        def __init__(self, name):
            self._m = {}

        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._m[o.__name__] = o
                    return o
                return deco
            self._m[obj.__name__] = obj
            return obj

        def get(self, name):
            return self._m[name]

    _mod("fvcore.common.registry").Registry = Registry

    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            return copy.deepcopy(self)

        def dump(self):          # yacs CfgNode.dump(): the config as a YAML string
            def plain(x):
                if isinstance(x, dict):
                    return {k: plain(v) for k, v in x.items()}
                if isinstance(x, (list, tuple)):
                    return [plain(v) for v in x]
                return x
            return yaml.safe_dump(plain(self))

        def _merge(self, d):
            for k, v in d.items():
                if isinstance(v, dict):
                    self[k]._merge(v)
                    continue
                if isinstance(v, str):
                    try:
                        v = ast.literal_eval(v)
                    except Exception:
                        pass
                self[k] = list(v) if isinstance(v, tuple) else v

        def merge_from_file(self, f):
            self._merge(yaml.safe_load(open(f)))

        def merge_from_list(self, lst):
            for k, v in zip(lst[0::2], lst[1::2]):
                node = self
                parts = k.split(".")
                for p in parts[:-1]:
                    node = node[p]
                try:
                    v = ast.literal_eval(v) if isinstance(v, str) else v
                except Exception:
                    pass
                node[parts[-1]] = v

    _mod("fvcore.common.config").CfgNode = CfgNode

    class _PM:
        def open(self, *a, **k):
            return open(*a, **k)

        def exists(self, p):
            return os.path.exists(p)

    _mod("iopath.common.file_io").PathManagerFactory = type("PMF", (), {"get": staticmethod(lambda key=None: _PM())})
    _mod("fairscale.nn.checkpoint").checkpoint_wrapper = lambda m, *a, **k: m
    torch.Tensor.cuda = lambda self, *a, **k: self        # EgoNCE hard-codes .cuda() (losses.py:158)
    sys.path.insert(0, "/root/reference")


install_stubs()
torch.set_num_threads(1)   # canonical goldens (1 vs 8 threads differ by 2e-7)

from oracle import csts_oracle as O                                      # noqa: E402
from slowfast.config.defaults import get_cfg                             # noqa: E402
from slowfast.models.custom_multimodal_builder import CSTS               # noqa: E402
from slowfast.models import attention as ref_attn                       # noqa: E402
from slowfast.models import av_attention as ref_av                      # noqa: E402
from slowfast.models import stem_helper as ref_stem                     # noqa: E402
from slowfast.models import losses as ref_losses                        # noqa: E402
from slowfast.models import common as ref_common                        # noqa: E402
from slowfast.utils import utils as ref_utils                           # noqa: E402

YAML = "/root/reference/configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"


def make_cfg(num_frames=8, extra=()):
    cfg = get_cfg()
    cfg.merge_from_file(YAML)
    cfg.NUM_GPUS = 0
    cfg.MODEL.LOSS_FUNC = "kldiv+egonce"
    cfg.DATA.NUM_FRAMES = num_frames
    for k, v in extra:
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def load_seeded(module, prefix="", seed=0):
    sd = {k: O.seeded_tensor(prefix + k, tuple(v.shape), seed) for k, v in module.state_dict().items()}
    module.load_state_dict(sd, strict=True)
    return module


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: np.asarray(v) for k, v in arrs.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


def t2n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ manifests
def gen_manifests():
    for T in (8, 16, 32):
        m = CSTS(make_cfg(T))
        man = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        with open(os.path.join(OUT, f"manifest_T{T}.json"), "w") as f:
            json.dump({"num_params": sum(p.numel() for p in m.parameters()), "entries": man}, f)
        print("manifest", T, len(man))
        if T == 8:
            # also the kldiv-only variant (no vision_proj/audio_proj)
            cfg = make_cfg(8)
            cfg.MODEL.LOSS_FUNC = "kldiv"
            m2 = CSTS(cfg)
            with open(os.path.join(OUT, "manifest_T8_kldiv.json"), "w") as f:
                json.dump({"num_params": sum(p.numel() for p in m2.parameters()),
                           "entries": [[k, list(v.shape)] for k, v in m2.state_dict().items()]}, f)
            geo = {"pool_kv_stride": cfg.MVIT.POOL_KV_STRIDE,
                   "blocks": [[b.dim, b.dim_out, b.attn.num_heads] for b in m.blocks]}
            with open(os.path.join(OUT, "geometry_T8.json"), "w") as f:
                json.dump(geo, f)


# ------------------------------------------------------------------ blocks
from functools import partial                                            # noqa: E402
LN6 = partial(torch.nn.LayerNorm, eps=1e-6)


def gen_blocks():
    g = torch.Generator().manual_seed(11)
    # config 1: 1x3x8x56x56 -> PatchEmbed -> (1,784,96), thw (4,14,14); blocks.0 and blocks.1 geometry
    clip = torch.randn(1, 3, 8, 56, 56, generator=g)
    pe = load_seeded(ref_stem.PatchEmbed(3, 96, (3, 7, 7), (2, 4, 4), (1, 3, 3)), "cfg1.patch_embed.")
    tok = pe(clip)
    b0 = load_seeded(ref_attn.MultiScaleBlock(96, 192, 1, 4.0, True, norm_layer=LN6, kernel_q=[], kernel_kv=[3, 3, 3],
                                              stride_q=[], stride_kv=[1, 8, 8], mode="conv", has_cls_embed=False),
                     "cfg1.b0.").eval()
    y0, thw0 = b0(tok, [4, 14, 14])
    b1 = load_seeded(ref_attn.MultiScaleBlock(192, 192, 2, 4.0, True, norm_layer=LN6, kernel_q=[3, 3, 3],
                                              kernel_kv=[3, 3, 3], stride_q=[1, 2, 2], stride_kv=[1, 4, 4],
                                              mode="conv", has_cls_embed=False), "cfg1.b1.").eval()
    y1, thw1 = b1(y0, thw0)
    save("block_cfg1.npz", clip=t2n(clip), tok=t2n(tok), y0=t2n(y0), thw0=thw0, y1=t2n(y1), thw1=thw1)

    # decoder blocks, both stride kinds
    x = torch.randn(2, 2 * 4 * 4, 192, generator=g)
    d1 = load_seeded(ref_attn.MultiScaleDecoderBlock(192, 96, 2, 4.0, True, norm_layer=LN6, kernel_q=[3, 3, 3],
                                                     kernel_kv=[3, 3, 3], stride_q=[1, 2, 2], stride_kv=[1, 2, 2],
                                                     mode="conv", has_cls_embed=False), "dec_a.").eval()
    ya, thwa = d1(x, [2, 4, 4])
    d2 = load_seeded(ref_attn.MultiScaleDecoderBlock(192, 96, 2, 4.0, True, norm_layer=LN6, kernel_q=[3, 3, 3],
                                                     kernel_kv=[3, 3, 3], stride_q=[2, 1, 1], stride_kv=[1, 4, 4],
                                                     mode="conv", has_cls_embed=False), "dec_b.").eval()
    yb, thwb = d2(x, [2, 4, 4])
    # head_dim 192 variant (decode_block2 geometry: 4 heads of 192 at dim 768 is too big; use dim 384, 2 heads)
    x3 = torch.randn(1, 2 * 2 * 2, 384, generator=g)
    d3 = load_seeded(ref_attn.MultiScaleDecoderBlock(384, 192, 2, 4.0, True, norm_layer=LN6, kernel_q=[3, 3, 3],
                                                     kernel_kv=[3, 3, 3], stride_q=[1, 2, 2], stride_kv=[1, 2, 2],
                                                     mode="conv", has_cls_embed=False), "dec_c.").eval()
    yc, thwc = d3(x3, [2, 2, 2])
    save("block_decoder.npz", x=t2n(x), ya=t2n(ya), thwa=thwa, yb=t2n(yb), thwb=thwb, x3=t2n(x3), yc=t2n(yc),
         thwc=thwc)

    # spatial / temporal fusion blocks
    xs = torch.randn(2, 2 * 2 * 2 + 2, 192, generator=g)        # T=2, H=W=2 -> 8 video + 2 audio tokens
    kw = dict(mlp_ratio=4.0, qkv_bias=True, norm_layer=LN6, kernel_q=[1, 1, 1], kernel_kv=[1, 1, 1],
              stride_q=[1, 1, 1], stride_kv=[1, 1, 1], mode="conv", has_cls_embed=False)
    sp = load_seeded(ref_av.SpatialBlock(192, 192, 2, **kw), "sp.").eval()
    ys, _ = sp(xs, [2, 2, 2])
    _, _, attn_s = sp(xs, [2, 2, 2], return_spatial_attn=True)
    sp2 = load_seeded(ref_av.SpatialBlock(192, 192, 2, return_audio_attn=True, **kw), "sp.").eval()
    ys2, _, aa = sp2(xs, [2, 2, 2])
    xt = torch.randn(2, 4, 192, generator=g)
    tp = load_seeded(ref_av.TemporalBlock(192, 192, 2, **kw), "tp.").eval()
    yt, _, attn_t = tp(xt, (2, 2, 2), return_temporal_attn=True)
    save("block_fusion.npz", xs=t2n(xs), ys=t2n(ys), attn_s=t2n(attn_s), ys2=t2n(ys2), audio_attn=t2n(aa),
         xt=t2n(xt), yt=t2n(yt), attn_t=t2n(attn_t))

    # losses
    logits = torch.randn(3, 1, 4, 8, 8, generator=g) * 3
    tgt = torch.rand(3, 4, 8, 8, generator=g)
    tgt = tgt / tgt.sum(dim=(-1, -2), keepdim=True)
    p = ref_utils.frame_softmax(logits, temperature=2)
    kl = ref_losses.KLDiv()(p, tgt)
    a = torch.randn(3, 16, generator=g)
    b = torch.randn(3, 16, generator=g)
    sim = ref_utils.sim_matrix(a, b)
    nce = ref_losses.EgoNCE()(sim)
    save("losses.npz", logits=t2n(logits), tgt=t2n(tgt), p=t2n(p), kl=t2n(kl), a=t2n(a), b=t2n(b), sim=t2n(sim),
         nce=t2n(nce))


# ------------------------------------------------------------------ full model
GRAD_NAMES = [
    "pos_embed_spatial", "pos_embed_temporal_audio", "patch_embed.proj.weight", "patch_embed_audio.proj.bias",
    "blocks.0.attn.qkv.weight", "blocks.0.attn.pool_k.weight", "blocks.1.attn.pool_q.weight",
    "blocks.1.attn.norm_q.weight", "blocks.3.mlp.fc1.weight", "blocks.7.attn.proj.bias", "blocks.13.proj.weight",
    "blocks.15.norm2.bias", "blocks_audio.1.attn.norm_v.bias", "blocks_audio.3.mlp.fc2.weight",
    "vision_pool.weight", "audio_pool.weight", "audio_pool2.bias", "temporal_fusion.attn.qkv.weight",
    "spatial_fusion.mlp.fc2.bias", "decode_block1.attn.upsample_q.weight", "decode_block2.attn.norm_q.weight",
    "decode_block3.proj.weight", "decode_block4.attn.pool_v.weight", "decode_block4.mlp.fc2.weight",
    "vision_proj.weight", "audio_proj.bias", "classifier.weight", "classifier.bias",
]


def gen_model():
    cfg = make_cfg(8)
    m = load_seeded(CSTS(cfg)).eval()
    batch = O.synthetic_batch(2, 8, 256, seed=1000)
    taps = {}
    hooks = [m.blocks[15].register_forward_hook(lambda mod, i, o: taps.__setitem__("enc_video", o[0])),
             m.blocks_audio[3].register_forward_hook(lambda mod, i, o: taps.__setitem__("enc_audio", o[0])),
             m.decode_block1.register_forward_hook(lambda mod, i, o: taps.__setitem__("x_reweight", i[0]))]
    logits, v, a = m([batch["video"]], batch["audio"], return_embed=True)
    for h in hooks:
        h.remove()
    p = ref_utils.frame_softmax(logits, temperature=2)
    sim = ref_utils.sim_matrix(v, a)
    kld = ref_losses.KLDiv()(p, batch["labels_hm"])
    nce = ref_losses.EgoNCE()(sim)
    loss = kld + 0.05 * nce
    m.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    gnorm = {n: float(named[n].grad.double().norm()) for n in GRAD_NAMES}   # float64: fp32 accumulation over 37.7 M elements is off by 3e-3
    total = float(torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in m.parameters())))
    # small gradient slices for elementwise comparison
    gslice = {n.replace(".", "_") + "_g": t2n(named[n].grad.flatten()[:64]) for n in GRAD_NAMES}
    argmax = p.reshape(2, 8, -1).argmax(-1)
    save("model_T8_B2.npz", logits=t2n(logits), v_emb=t2n(v), a_emb=t2n(a), heat=t2n(p).astype(np.float32),
         argmax=t2n(argmax), kld=t2n(kld), nce=t2n(nce), loss=t2n(loss),
         enc_video=t2n(taps["enc_video"])[:, :, :64], enc_audio=t2n(taps["enc_audio"])[:, :, :64],
         x_reweight=t2n(taps["x_reweight"])[:, :, :64],
         grad_names=np.array(GRAD_NAMES), grad_norms=np.array([gnorm[n] for n in GRAD_NAMES]),
         grad_total_norm=total, **gslice)

    # drop-path (train mode) golden: record the reference's torch.rand draws
    m.train()
    rec = []
    orig_rand = torch.rand

    def rand_rec(*a_, **k_):
        r = orig_rand(*a_, **k_)
        rec.append(r.flatten().clone())
        return r
    torch.manual_seed(4242)
    torch.rand = rand_rec
    try:
        with torch.no_grad():
            logits_tr = m([batch["video"]], batch["audio"])
    finally:
        torch.rand = orig_rand
    save("model_T8_B2_droppath.npz", logits=t2n(logits_tr), rand=np.stack([t2n(r) for r in rec]))
    m.eval()

    # spatial-audio-attn variant + returned attention maps (forward only, B=1)
    with torch.no_grad():
        b1 = O.synthetic_batch(1, 8, 256, seed=1001)
        out = m([b1["video"]], b1["audio"], return_spatial_attn=True, return_temporal_attn=True)
        save("model_T8_B1_attn.npz", logits=t2n(out[0]), spatial_attn=t2n(out[1]).astype(np.float16),
             temporal_attn=t2n(out[2]))
        cfg2 = make_cfg(8, [("MVIT.SPATIAL_AUDIO_ATTN", True)])
        m2 = load_seeded(CSTS(cfg2)).eval()
        lg2 = m2([b1["video"]], b1["audio"])
        save("model_T8_B1_saa.npz", logits=t2n(lg2))
        del m2

    # T=16 forward, B=1
    del m
    cfg16 = make_cfg(16)
    m16 = load_seeded(CSTS(cfg16)).eval()
    b16 = O.synthetic_batch(1, 16, 256, seed=1002)
    with torch.no_grad():
        lg, v16, a16 = m16([b16["video"]], b16["audio"], return_embed=True)
    save("model_T16_B1.npz", logits=t2n(lg), v_emb=t2n(v16), a_emb=t2n(a16))


def gen_model_t32():
    """BASELINE config 5 geometry: the Aria YAML with DATA.NUM_FRAMES 32 (the longer-T token grid), B = 1 forward."""
    from slowfast.models.custom_multimodal_builder import CSTS
    cfg = get_cfg()
    cfg.merge_from_file("/root/reference/configs/Aria/CSTS_Aria_Gaze_Forecast.yaml")
    cfg.NUM_GPUS = 0
    cfg.MODEL.LOSS_FUNC = "kldiv+egonce"
    cfg.DATA.NUM_FRAMES = 32
    m = load_seeded(CSTS(cfg)).eval()
    b = O.synthetic_batch(1, 32, 256, seed=1003)
    with torch.no_grad():
        lg, v, a = m([b["video"]], b["audio"], return_embed=True)
    heat = ref_utils.frame_softmax(lg, 2)
    save("model_T32_B1_aria.npz", logits=t2n(lg).astype(np.float16), v_emb=t2n(v), a_emb=t2n(a),
         argmax=t2n(heat).reshape(32, -1).argmax(-1).astype(np.int32),
         logits_head=t2n(lg).reshape(-1)[:4096].astype(np.float32))


def _train_fixture(m, batch, names, alpha=0.05):
    """Eval-mode (no drop-path) forward + KLDiv + alpha * EgoNCE + backward of the reference model: the quantities the
    GPU train-step parity tests compare (tools/train_avgaze_net.py:70-99)."""
    logits, v, a = m([batch["video"]], batch["audio"], return_embed=True)
    p = ref_utils.frame_softmax(logits, temperature=2)
    kld = ref_losses.KLDiv()(p, batch["labels_hm"])
    nce = ref_losses.EgoNCE()(ref_utils.sim_matrix(v, a))
    loss = kld + alpha * nce
    m.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    gnorm = [float(named[n].grad.double().norm()) for n in names]
    total = float(torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in m.parameters() if p_.grad is not None)))
    gslice = {n.replace(".", "_") + "_g": t2n(named[n].grad.flatten()[:64]) for n in names}
    B, T = p.shape[0], p.shape[2]
    out = dict(logits=t2n(logits).astype(np.float16), logits_head=t2n(logits).reshape(-1)[:4096].astype(np.float32),
               v_emb=t2n(v), a_emb=t2n(a), argmax=t2n(p.reshape(B, T, -1).argmax(-1)).astype(np.int32),
               heat_head=t2n(p).reshape(-1)[:4096].astype(np.float32), kld=t2n(kld), nce=t2n(nce), loss=t2n(loss),
               grad_names=np.array(names), grad_norms=np.array(gnorm), grad_total_norm=total, **gslice)
    return out


def gen_train_t16():
    """The BENCHMARKED token grid (16 x 256^2): B = 2 train fixture (loss, KLDiv, EgoNCE, 28 gradient norms + slices,
    total norm), same recipe as gen_model."""
    m = load_seeded(CSTS(make_cfg(16))).eval()
    batch = O.synthetic_batch(2, 16, 256, seed=1004)
    save("model_T16_B2_train.npz", **_train_fixture(m, batch, GRAD_NAMES))


T32_GRAD_NAMES = [
    "pos_embed_temporal", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.1.attn.pool_k.weight",
    "blocks.3.mlp.fc1.weight", "blocks.14.attn.proj.weight", "blocks_audio.2.attn.norm_k.weight", "vision_pool.weight",
    "spatial_fusion.attn.qkv.weight", "decode_block2.attn.upsample_q.weight", "decode_block4.mlp.fc2.weight",
    "classifier.weight",
]


def gen_train_t32():
    """BASELINE config 5 geometry (Aria YAML + DATA.NUM_FRAMES 32): B = 1 forward + backward (EgoNCE is identically 0 at
    B = 1): loss + 12 gradient norms + slices -- the path with multi-tile keys (N_kv = 4096)."""
    cfg = get_cfg()
    cfg.merge_from_file("/root/reference/configs/Aria/CSTS_Aria_Gaze_Forecast.yaml")
    cfg.NUM_GPUS = 0
    cfg.MODEL.LOSS_FUNC = "kldiv+egonce"
    cfg.DATA.NUM_FRAMES = 32
    m = load_seeded(CSTS(cfg)).eval()
    batch = O.synthetic_batch(1, 32, 256, seed=1003)
    save("model_T32_B1_aria_train.npz", **_train_fixture(m, batch, T32_GRAD_NAMES))


SAA_GRAD_NAMES = ["spatial_fusion.attn.qkv.weight", "spatial_fusion.attn.qkv.bias", "spatial_fusion.norm1.weight",
                  "audio_pool.weight", "vision_pool.weight", "blocks.15.mlp.fc2.weight", "blocks_audio.3.mlp.fc2.weight",
                  "decode_block1.attn.qkv.weight", "classifier.weight"]


def gen_train_saa():
    """MVIT.SPATIAL_AUDIO_ATTN True in TRAIN use: the gradient that flows through the min-max-rescaled audio->pixel
    attention map (av_attention.py:356-370 -> custom_multimodal_builder.py:438-440), T = 8, B = 2."""
    cfg = make_cfg(8, [("MVIT.SPATIAL_AUDIO_ATTN", True)])
    m = load_seeded(CSTS(cfg)).eval()
    batch = O.synthetic_batch(2, 8, 256, seed=1005)
    save("model_T8_B2_saa_train.npz", **_train_fixture(m, batch, SAA_GRAD_NAMES))


def bench_fingerprint(model, batch):
    """A few float64 sums that identify (weights, batch) of bench.py's loss check (tolerant comparison: see bench.py)."""
    sd = model.state_dict()
    names = ["pos_embed_spatial", "blocks.0.attn.qkv.weight", "blocks.15.mlp.fc2.weight", "vision_pool.weight",
             "decode_block4.mlp.fc2.weight", "classifier.weight"]
    fp = {n: float(sd[n].double().abs().sum()) for n in names}
    for k in ("video", "audio", "labels_hm"):
        fp["batch." + k] = float(batch[k].double().abs().sum())
    return fp


def gen_bench_expected():
    """Expected loss of bench.py's loss check: the BENCHMARKED configuration (16 x 256^2, b = 4, kldiv+egonce) in eval
    mode on the CPU-generated seed-1000 batch, computed by the imported reference with the weights csts_amd's own model
    constructor draws under torch.manual_seed(cfg.RNG_SEED) -- exactly what bench.py builds on the GPU box."""
    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    from csts_amd import train as T
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    out = {}
    for frames, b in ((16, 4),):
        cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"),
                        ["NUM_GPUS", 0, "TRAIN.BATCH_SIZE", b, "MODEL.LOSS_FUNC", "kldiv+egonce", "MODEL.LOSS_ALPHA", 0.05,
                         "DATA.NUM_FRAMES", frames, "CSTS_AMD.COMPUTE", "bf16"])
        torch.manual_seed(cfg.RNG_SEED)
        mine = build_model(cfg)
        batch = T.synthetic_batch(b, frames, 256, 1000, "cpu", pipeline="torch")
        ref = CSTS(make_cfg(frames)).eval()
        ref.load_state_dict(mine.state_dict(), strict=True)
        with torch.no_grad():
            logits, v, a = ref([batch["video"]], batch["audio"], return_embed=True)
            p = ref_utils.frame_softmax(logits, temperature=2)
            kld = ref_losses.KLDiv()(p, batch["labels_hm"])
            nce = ref_losses.EgoNCE()(ref_utils.sim_matrix(v, a))
        out[f"T{frames}_B{b}"] = {"loss": float(kld + 0.05 * nce), "kld": float(kld), "nce": float(nce),
                                  "fingerprint": bench_fingerprint(mine, batch), "seed": 1000, "rng_seed": int(cfg.RNG_SEED)}
        print(out)
    with open(os.path.join(OUT, "bench_expected.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_metrics():
    """adaptive_f1 of the reference (slowfast/utils/metrics.py) on seeded heat maps, all three threshold tables."""
    from slowfast.utils import metrics as ref_metrics
    g = torch.Generator().manual_seed(4321)
    B, T = 3, 8
    logits = torch.randn(B, 1, T, 64, 64, generator=g) * 2.0
    batch = O.synthetic_batch(B, T, 256, seed=55)
    hm = batch["labels_hm"]
    # pull the prediction towards the label so that the sweep has a non-trivial optimum
    logits = logits + 40.0 * hm.unsqueeze(1) / hm.amax(dim=(-1, -2), keepdim=True).unsqueeze(1)
    preds = ref_utils.frame_softmax(logits, 2)
    flat = preds.view(preds.size()[:-2] + (preds.size(-1) * preds.size(-2),))
    resc = ((flat - flat.min(dim=-1, keepdim=True)[0]) / (flat.max(dim=-1, keepdim=True)[0] - flat.min(dim=-1, keepdim=True)[0] + 1e-6)).view(preds.size())
    labels = batch["labels"].clone()
    labels[0, 2, 2] = 1.0          # untracked frames (type != fixation) are skipped
    labels[2, 5, 2] = 2.0
    out = {}
    for ds in ("ego4d_av_gaze_forecast", "aria_av_gaze_forecast", "ego4d_av_gaze"):
        out[ds] = np.array(ref_metrics.adaptive_f1(resc, hm, labels, dataset=ds), dtype=np.float64)
    save("metrics_f1.npz", logits=t2n(logits).astype(np.float32), labels=t2n(labels), **out)


def gen_checkpoint():
    """A .pyth checkpoint written by the reference's own save_checkpoint (slowfast/utils/checkpoint.py:110-143) for a
    small stand-in module with CSTS-style names (incl. a pos_embed_temporal of a different length) and a torch AdamW
    state -- the wire format csts_amd.checkpoint must read -- plus what the reference's load_checkpoint makes of it."""
    import tempfile, shutil
    from slowfast.utils import checkpoint as ref_ckpt

    class PM:          # the iopath stub lacks mkdirs
        def mkdirs(self, p): os.makedirs(p, exist_ok=True)
        def open(self, *a, **k): return open(*a, **k)
        def exists(self, p): return os.path.exists(p)
        def ls(self, p): return os.listdir(p)
    ref_ckpt.pathmgr = PM()

    class Tiny(torch.nn.Module):
        def __init__(self, T):
            super().__init__()
            self.pos_embed_spatial = torch.nn.Parameter(torch.zeros(1, 16, 8))
            self.pos_embed_temporal = torch.nn.Parameter(torch.zeros(1, T, 8))
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(2)])
            self.head = torch.nn.Linear(8, 3)

    torch.manual_seed(11)
    src = Tiny(4)
    for p_ in src.parameters():
        torch.nn.init.normal_(p_, std=0.5)
    opt = torch.optim.AdamW(src.parameters(), lr=1e-3, eps=1e-8, weight_decay=0.05)
    src.head(src.blocks[1](src.blocks[0](src.pos_embed_spatial + src.pos_embed_temporal.mean(1, keepdim=True)))).sum().backward()
    opt.step()
    cfg = make_cfg(8)
    tmp = tempfile.mkdtemp()
    path = ref_ckpt.save_checkpoint(tmp + "/", src, opt, 6, cfg)
    assert path.endswith("checkpoint_epoch_00007.pyth"), path
    shutil.copy(path, os.path.join(OUT, "ref_checkpoint_epoch_00007.pyth"))
    # the reference loading it into a model whose temporal pos-embed is longer and whose head has another shape
    dst = Tiny(8)
    dst.head = torch.nn.Linear(8, 5)
    torch.manual_seed(12)
    for p_ in dst.parameters():
        torch.nn.init.normal_(p_, std=0.1)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    epoch = ref_ckpt.load_checkpoint(path, dst, data_parallel=False, optimizer=None, epoch_reset=True)
    after = dst.state_dict()
    save("ref_checkpoint_loaded.npz", epoch=np.array(epoch), **{k.replace(".", "__"): t2n(v) for k, v in after.items()},
         **{"before__" + k.replace(".", "__"): t2n(v) for k, v in before.items()})
    shutil.rmtree(tmp)


def gen_checkpoint_pair():
    """The video + audio pre-training pair (slowfast/utils/checkpoint.py:357-470; TRAIN.CHECKPOINT_FILE_PATH +
    TRAIN.AUDIO_CHECKPOINT_FILE_PATH, :645-656): two .pyth files written by the reference's save_checkpoint for small stand-in
    modules with CSTS-style names, and what the reference's load_video_and_audio_checkpoints makes of them in a model whose
    position embeddings have other lengths (bilinear resize of both modalities) and whose head has another shape."""
    import tempfile, shutil
    from slowfast.utils import checkpoint as ref_ckpt

    class PM:
        def mkdirs(self, p): os.makedirs(p, exist_ok=True)
        def open(self, *a, **k): return open(*a, **k)
        def exists(self, p): return os.path.exists(p)
        def ls(self, p): return os.listdir(p)
    ref_ckpt.pathmgr = PM()

    class Tiny(torch.nn.Module):
        def __init__(self, T, S, nout, audio=True, video=True):
            super().__init__()
            if video:
                self.pos_embed_spatial = torch.nn.Parameter(torch.zeros(1, S, 8))
                self.pos_embed_temporal = torch.nn.Parameter(torch.zeros(1, T, 8))
                self.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(2)])
            if audio:
                self.pos_embed_spatial_audio = torch.nn.Parameter(torch.zeros(1, S, 8))
                self.pos_embed_temporal_audio = torch.nn.Parameter(torch.zeros(1, T, 8))
                self.blocks_audio = torch.nn.ModuleList([torch.nn.Linear(8, 8)])
            self.head = torch.nn.Linear(8, nout)

    cfg = make_cfg(8)
    tmp = tempfile.mkdtemp()
    paths = {}
    for name, seed, kw, ep in (("video", 21, dict(T=4, S=16, nout=3, audio=False), 4), ("audio", 22, dict(T=2, S=9, nout=3, video=False), 9)):
        torch.manual_seed(seed)
        src = Tiny(**kw)
        for p_ in src.parameters():
            torch.nn.init.normal_(p_, std=0.5)
        opt = torch.optim.AdamW(src.parameters(), lr=1e-3)
        d = os.path.join(tmp, name) + "/"
        path = ref_ckpt.save_checkpoint(d, src, opt, ep, cfg)
        paths[name] = os.path.join(OUT, f"ref_pretrain_{name}.pyth")
        shutil.copy(path, paths[name])
    dst = Tiny(T=8, S=16, nout=5)
    torch.manual_seed(23)
    for p_ in dst.parameters():
        torch.nn.init.normal_(p_, std=0.1)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    epoch = ref_ckpt.load_video_and_audio_checkpoints(paths["video"], paths["audio"], dst, data_parallel=False, optimizer=None,
                                                      epoch_reset=True)
    dst2 = Tiny(T=8, S=16, nout=3)
    dst2.load_state_dict({k: v.clone() for k, v in before.items() if not k.startswith("head")}, strict=False)
    epoch2 = ref_ckpt.load_video_and_audio_checkpoints(paths["video"], paths["audio"], dst2, data_parallel=False, optimizer=None,
                                                       epoch_reset=False)
    save("ref_pretrain_pair_loaded.npz", epoch=np.array(epoch), epoch_no_reset=np.array(epoch2),
         **{k.replace(".", "__"): t2n(v) for k, v in dst.state_dict().items()},
         **{"before__" + k.replace(".", "__"): t2n(v) for k, v in before.items()},
         **{"noreset__" + k.replace(".", "__"): t2n(v) for k, v in dst2.state_dict().items()})
    shutil.rmtree(tmp)


def _autocast_fixture(m, batch, names, dtype, alpha=0.05):
    """The reference's OWN mixed-precision error on this path: the same model, weights and batch once in fp32 and once under
    ``torch.autocast(dtype)`` exactly as the training loop wraps forward + losses (tools/train_avgaze_net.py:70-88; CPU
    autocast, the only one this container can run).  Stored: the autocast outputs (heat maps, arg-max) and the error of every
    quantity the GPU tests bar against the fp32 run -- these numbers are what "no worse than the reference's autocast" means."""
    def run(ctx, scale=1.0):
        with ctx:
            logits, v, a = m([batch["video"]], batch["audio"], return_embed=True)
            p = ref_utils.frame_softmax(logits, temperature=2)
            kld = ref_losses.KLDiv()(p, batch["labels_hm"])
            nce = ref_losses.EgoNCE()(ref_utils.sim_matrix(v, a))
            loss = kld + alpha * nce
        # fp16: GradScaler semantics (train_avgaze_net.py:99-109,277): backward of scale * loss (initial scale 2^16), unscale_,
        # and on an inf / nan gradient the step is skipped and the scale halved -- replayed here until the gradients are finite
        while True:
            m.zero_grad()
            (loss.float() * scale).backward(retain_graph=True)
            finite = all(bool(torch.isfinite(p_.grad).all()) for p_ in m.parameters() if p_.grad is not None)
            if finite or scale == 1.0:
                break
            scale *= 0.5
        named = dict(m.named_parameters())
        g = {n: named[n].grad.detach().double().clone() / scale for n in names}
        total = float(torch.sqrt(sum((p_.grad.double() ** 2).sum() for p_ in m.parameters() if p_.grad is not None))) / scale
        return dict(loss_scale=scale, logits=logits.detach().float(), heat=p.detach().float(), v=v.detach().float(), a=a.detach().float(),
                    kld=float(kld), nce=float(nce), loss=float(loss), g=g, total=total)
    import contextlib
    ref = run(contextlib.nullcontext())
    amp = run(torch.autocast("cpu", dtype=dtype), 65536.0 if dtype == torch.float16 else 1.0)

    def rel(x, y):
        return float((x.double() - y.double()).norm() / y.double().norm())
    B, T = ref["heat"].shape[0], ref["heat"].shape[2]
    am_ref = ref["heat"].reshape(B, T, -1).argmax(-1)
    am_amp = amp["heat"].reshape(B, T, -1).argmax(-1)
    cos = [float((amp["g"][n] * ref["g"][n]).sum() / (amp["g"][n].norm() * ref["g"][n].norm() + 1e-300)) for n in names]
    return dict(
        heat_amp=t2n(amp["heat"]).astype(np.float16), heat_ref=t2n(ref["heat"]).astype(np.float32),
        logits_ref=t2n(ref["logits"]).astype(np.float32), argmax_amp=t2n(am_amp).astype(np.int32), argmax_ref=t2n(am_ref).astype(np.int32),
        heat_rel_l2=rel(amp["heat"], ref["heat"]), logits_rel_l2=rel(amp["logits"], ref["logits"]),
        v_emb_rel_l2=rel(amp["v"], ref["v"]), a_emb_rel_l2=rel(amp["a"], ref["a"]),
        argmax_agree=float((am_ref == am_amp).float().mean()),
        loss_ref=ref["loss"], loss_amp=amp["loss"], kld_ref=ref["kld"], kld_amp=amp["kld"], nce_ref=ref["nce"], nce_amp=amp["nce"],
        grad_names=np.array(names),
        grad_norm_ref=np.array([float(ref["g"][n].norm()) for n in names]),
        grad_norm_amp=np.array([float(amp["g"][n].norm()) for n in names]),
        grad_rel_l2=np.array([rel(amp["g"][n], ref["g"][n]) for n in names]), grad_cos=np.array(cos),
        grad_total_norm_ref=ref["total"], grad_total_norm_amp=amp["total"], dtype=str(dtype), loss_scale=amp["loss_scale"])


def gen_autocast():
    """what the reference's mixed precision itself costs, at the fixture batches of the fp32
    goldens (same seeds), so that the bf16 compute mode is barred against the reference's autocast and not against itself."""
    torch.set_num_threads(8)
    m = load_seeded(CSTS(make_cfg(8))).eval()
    save("autocast_bf16_T8_B2.npz", **_autocast_fixture(m, O.synthetic_batch(2, 8, 256, seed=1000), GRAD_NAMES, torch.bfloat16))
    save("autocast_fp16_T8_B2.npz", **_autocast_fixture(m, O.synthetic_batch(2, 8, 256, seed=1000), GRAD_NAMES, torch.float16))
    del m
    m = load_seeded(CSTS(make_cfg(16))).eval()
    save("autocast_bf16_T16_B2.npz", **_autocast_fixture(m, O.synthetic_batch(2, 16, 256, seed=1004), GRAD_NAMES, torch.bfloat16))
    del m
    cfg = get_cfg()
    cfg.merge_from_file("/root/reference/configs/Aria/CSTS_Aria_Gaze_Forecast.yaml")
    cfg.NUM_GPUS = 0
    cfg.MODEL.LOSS_FUNC = "kldiv+egonce"
    cfg.DATA.NUM_FRAMES = 32
    m = load_seeded(CSTS(cfg)).eval()
    b = O.synthetic_batch(1, 32, 256, seed=1003)
    save("autocast_fp16_T32_B1_aria.npz", **_autocast_fixture(m, b, T32_GRAD_NAMES, torch.float16))
    save("autocast_bf16_T32_B1_aria.npz", **_autocast_fixture(m, b, T32_GRAD_NAMES, torch.bfloat16))


def gen_lr():
    from slowfast.utils import lr_policy
    cfg = make_cfg(8)
    pts = [0.0, 0.5, 1.0, 3.25, 7.5, 14.0, 14.99]
    save("lr_schedule.npz", epochs=np.array(pts), lr=np.array([lr_policy.get_lr_at_epoch(cfg, e) for e in pts]))


if __name__ == "__main__":
    what = sys.argv[1:] or ["manifest", "blocks", "model", "lr"]
    if "manifest" in what:
        gen_manifests()
    if "blocks" in what:
        gen_blocks()
    if "lr" in what:
        gen_lr()
    if "metrics" in what or "blocks" in what:
        gen_metrics()
    if "checkpoint" in what or "blocks" in what:
        gen_checkpoint()
    if "checkpoint_pair" in what or "blocks" in what:
        gen_checkpoint_pair()
    if "model" in what:
        gen_model()
    if "model" in what or "t32" in what:
        gen_model_t32()
    if "train16" in what:
        gen_train_t16()
    if "train32" in what:
        gen_train_t32()
    if "trainsaa" in what:
        gen_train_saa()
    if "bench" in what:
        gen_bench_expected()
    if "autocast" in what:
        gen_autocast()
