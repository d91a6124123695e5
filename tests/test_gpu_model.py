"""GPU parity tests, model level: the HIP CSTS model against (a) the golden fixtures captured from the reference and
(b) the CPU oracle on the same seeded weights/inputs.  Bars (BASELINE.json north_star): fp32 mode heatmaps within
1e-3 rel-L2 of the reference, per-frame argmax bit-exact; bf16 mode is reported against looser, stated tolerances
(the reference's own bf16 autocast on the fixture weights / batches deviates 8.4e-3 ... 9.8e-3 on logits and 3.4e-3 on
heatmaps from its fp32 run: tests/golden/autocast_bf16_*.npz; SURVEY.md D3's 7.6e-4 was measured on default-initialised
weights and Gaussian inputs)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2, GOLDEN

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from csts_amd import lib as L                      # noqa: E402
from csts_amd.config import load_yaml              # noqa: E402
from csts_amd.build import build_model             # noqa: E402
from csts_amd.model import Block, Runtime          # noqa: E402
from csts_amd import ops, train as T               # noqa: E402
from oracle import csts_oracle as O                # noqa: E402

DEV = torch.device("cuda:0")
ROOT = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
YAML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml")


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


_MODELS = {}


def make_model(compute="fp32", T_=8, opts=()):
    key = (compute, T_, tuple(opts))
    if key not in _MODELS:
        cfg = load_yaml(YAML, ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", T_,
                               "CSTS_AMD.COMPUTE", compute] + list(opts))
        m = build_model(cfg)
        m.load_state_dict(O.seeded_params(T_, 256), strict=True)
        m.eval()
        _MODELS.clear()          # keep one 188 M-parameter model alive at a time
        _MODELS[key] = (m, cfg)
    return _MODELS[key]


def dev_batch(B, T_, seed):
    b = O.synthetic_batch(B, T_, 256, seed=seed)
    return {k: v.to(DEV) for k, v in b.items()}


# ------------------------------------------------------------------------------------------------ blocks vs fixtures
def _block_from_spec(spec, seed_prefix, compute="fp32"):
    rt = Runtime(compute)
    blk = Block(spec.kind, spec.dim, spec.dim_out, spec.heads, spec.stride_q, spec.stride_kv, spec.has_pool_q,
                spec.has_pool_kv, spec.mlp_hidden, 0.0, rt)
    sd = {k: O.seeded_tensor(seed_prefix + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd, strict=True)
    return blk.to(DEV).eval()


@pytest.mark.parametrize("compute,tol", [("fp32", 3e-5), ("bf16", 3e-2)])
def test_block_cfg1_fixture(compute, tol):
    """BASELINE config 1 geometry (1x3x8x56x56 -> tokens (1,784,96)); blocks.0 and blocks.1 style blocks."""
    g = _load("block_cfg1.npz")
    rt = Runtime(compute)
    clip = torch.from_numpy(g["clip"]).to(DEV)
    w = O.seeded_tensor("cfg1.patch_embed.proj.weight", (96, 3, 3, 7, 7)).to(DEV)
    b = O.seeded_tensor("cfg1.patch_embed.proj.bias", (96,)).to(DEV)
    zs, zt = torch.zeros(1, 196, 96, device=DEV), torch.zeros(1, 4, 96, device=DEV)
    tok = ops.patch_embed(clip, w, b, zs, zt, (3, 7, 7), (2, 4, 4), (1, 3, 3), rt.act_dt, rt.compute)
    assert rel_l2(tok, g["tok"]) < tol
    s0 = O.BlockSpec("b0", "enc", 96, 192, 1, (1, 1, 1), (1, 8, 8), False, True, 384)
    s1 = O.BlockSpec("b1", "enc", 192, 192, 2, (1, 2, 2), (1, 4, 4), True, True, 768)
    with torch.no_grad():
        y0, thw0, _ = _block_from_spec(s0, "cfg1.b0.", compute)(torch.from_numpy(g["tok"]).to(DEV), [4, 14, 14])
        assert list(thw0) == list(g["thw0"]) and rel_l2(y0, g["y0"]) < tol
        y1, thw1, _ = _block_from_spec(s1, "cfg1.b1.", compute)(torch.from_numpy(g["y0"]).to(DEV), thw0)
        assert list(thw1) == list(g["thw1"]) and rel_l2(y1, g["y1"]) < tol


def test_block_decoder_and_fusion_fixtures():
    g = _load("block_decoder.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    with torch.no_grad():
        for key, pre, sq, skv, dim, dout, xin, thw in (
                ("a", "dec_a.", (1, 2, 2), (1, 2, 2), 192, 96, x, [2, 4, 4]),
                ("b", "dec_b.", (2, 1, 1), (1, 4, 4), 192, 96, x, [2, 4, 4]),
                ("c", "dec_c.", (1, 2, 2), (1, 2, 2), 384, 192, torch.from_numpy(g["x3"]).to(DEV), [2, 2, 2])):
            spec = O.BlockSpec("d", "dec", dim, dout, 2, sq, skv, True, True, 4 * dout)
            y, t2, _ = _block_from_spec(spec, pre)(xin, thw)
            assert list(t2) == list(g["thw" + key]) and rel_l2(y, g["y" + key]) < 3e-5, key
        f = _load("block_fusion.npz")
        ss = O.BlockSpec("s", "spatial", 192, 192, 2, (1, 1, 1), (1, 1, 1), False, False, 768)
        blk = _block_from_spec(ss, "sp.")
        xs = torch.from_numpy(f["xs"]).to(DEV)
        ys, _, attn = blk(xs, [2, 2, 2], want_attn=True)
        assert rel_l2(ys, f["ys"]) < 3e-5 and rel_l2(attn, f["attn_s"]) < 1e-4
        ys2, _, (aa, wmap) = blk(xs, [2, 2, 2], spatial_audio_attn=True)
        assert rel_l2(ys2, f["ys2"]) < 3e-5 and rel_l2(aa, f["audio_attn"]) < 1e-3
        assert rel_l2(wmap, f["audio_attn"].mean(axis=1).reshape(2, -1)) < 1e-3
        st = O.BlockSpec("t", "temporal", 192, 192, 2, (1, 1, 1), (1, 1, 1), False, False, 768)
        yt, _, at = _block_from_spec(st, "tp.")(torch.from_numpy(f["xt"]).to(DEV), (2, 2, 2), want_attn=True)
        assert rel_l2(yt, f["yt"]) < 3e-5 and rel_l2(at, f["attn_t"]) < 1e-4


# ------------------------------------------------------------------------------------------------ full model
def test_state_dict_surface():
    m, cfg = make_model("fp32")
    import json
    ref = json.load(open(os.path.join(GOLDEN, "manifest_T8.json")))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["entries"]
    assert all(p.is_cuda for p in m.parameters())


def test_full_model_fp32_vs_reference_golden():
    """THE parity gate: fp32 mode vs outputs of the imported reference (8x256^2, B=2): forward, losses, gradients."""
    m, cfg = make_model("fp32")
    g = _load("model_T8_B2.npz")
    batch = dev_batch(2, 8, 1000)
    loss, kld, nce, preds = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
    for p in m.parameters():
        p.grad = None
    loss.backward()
    heat = preds.detach()
    assert rel_l2(heat, g["heat"]) < 1e-3                      # north_star bar; expected ~1e-5
    assert rel_l2(heat, g["heat"]) < 1e-4
    assert (heat.reshape(2, 8, -1).argmax(-1).cpu().numpy() == g["argmax"]).all()      # index bit-exact
    assert abs(float(kld) - float(g["kld"])) < 1e-4 and abs(float(nce) - float(g["nce"])) < 1e-3
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    named = dict(m.named_parameters())
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if n == "classifier.bias":
            continue
        gr = named[n].grad
        gnorm = float(gr.double().norm())
        assert abs(gnorm - ref_norm) <= 2e-3 * ref_norm, (n, gnorm, ref_norm)
        assert rel_l2(gr.flatten()[:64], g[n.replace(".", "_") + "_g"]) < 5e-3, n
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    assert abs(total - float(g["grad_total_norm"])) < 1e-3 * float(g["grad_total_norm"])


def test_full_model_fp32_logits_embeddings_taps():
    m, cfg = make_model("fp32")
    g = _load("model_T8_B2.npz")
    batch = dev_batch(2, 8, 1000)
    with torch.no_grad():
        logits, v, a = m([batch["video"]], batch["audio"], return_embed=True)
    assert logits.shape == (2, 1, 8, 64, 64)
    assert rel_l2(logits, g["logits"]) < 1e-4 and rel_l2(v, g["v_emb"]) < 1e-4 and rel_l2(a, g["a_emb"]) < 1e-4


def test_full_model_bf16_mode_reported_tolerances():
    m, cfg = make_model("bf16")
    g = _load("model_T8_B2.npz")
    batch = dev_batch(2, 8, 1000)
    with torch.no_grad():
        logits, v, a = m([batch["video"]], batch["audio"], return_embed=True)
        heat = ops.frame_softmax(logits, 2.0)
    e_logits, e_heat = rel_l2(logits, g["logits"]), rel_l2(heat, g["heat"])
    agree = float((heat.reshape(2, 8, -1).argmax(-1).cpu().numpy() == g["argmax"]).mean())
    print(f"\n[bf16 mode] logits rel-L2 {e_logits:.3e}  heatmap rel-L2 {e_heat:.3e}  argmax agreement {agree:.3f}")
    # measured 4.7e-3 / 1.7e-3 / 1.0; the reference's own bf16 autocast on these weights and this batch: 9.8e-3 / 3.4e-3 / 1.0
    # (tests/golden/autocast_bf16_T8_B2.npz; test_bf16_mode_no_worse_than_reference_autocast holds the mode to those)
    assert e_logits < 6e-3 and e_heat < 2e-3
    assert agree == 1.0


def test_droppath_train_mode_fp32():
    m, cfg = make_model("fp32")
    g = _load("model_T8_B2_droppath.npz")
    rnd = torch.from_numpy(g["rand"])
    G = O.derive_geometry()
    active = [s for s in G["video"] if s.drop_path > 0]
    km = {}
    for i, s in enumerate(active):
        keep = 1.0 - s.drop_path
        km[s.prefix] = (torch.floor(keep + rnd[2 * i]), torch.floor(keep + rnd[2 * i + 1]))
    batch = dev_batch(2, 8, 1000)
    with torch.no_grad():
        logits = m([batch["video"]], batch["audio"], keep_masks=km)
    assert rel_l2(logits, g["logits"]) < 1e-4


def test_attention_outputs_fp32():
    m, cfg = make_model("fp32")
    g = _load("model_T8_B1_attn.npz")
    b1 = dev_batch(1, 8, 1001)
    with torch.no_grad():
        out = m([b1["video"]], b1["audio"], return_spatial_attn=True, return_temporal_attn=True)
    assert rel_l2(out[0], g["logits"]) < 1e-4
    assert rel_l2(out[1], g["spatial_attn"].astype(np.float32)) < 3e-3      # fixture stored as fp16
    assert rel_l2(out[2], g["temporal_attn"]) < 1e-4


def test_spatial_audio_attn_variant_fp32():
    m, cfg = make_model("fp32", 8, ("MVIT.SPATIAL_AUDIO_ATTN", True))
    b1 = dev_batch(1, 8, 1001)
    with torch.no_grad():
        lg = m([b1["video"]], b1["audio"])
    assert rel_l2(lg, _load("model_T8_B1_saa.npz")["logits"]) < 1e-4


def test_T16_forward_fp32():
    m, cfg = make_model("fp32", 16)
    g = _load("model_T16_B1.npz")
    b = dev_batch(1, 16, 1002)
    with torch.no_grad():
        lg, v, a = m([b["video"]], b["audio"], return_embed=True)
    assert lg.shape == (1, 1, 16, 64, 64)
    assert rel_l2(lg, g["logits"]) < 1e-4 and rel_l2(v, g["v_emb"]) < 1e-4 and rel_l2(a, g["a_emb"]) < 1e-4


def test_T32_aria_forward_fp32_and_bf16():
    """BASELINE config 5 geometry: CSTS_Aria_Gaze_Forecast.yaml with DATA.NUM_FRAMES 32 (4x the token grid of the YAML
    default: K/V of 4096 tokens per head exceed LDS, the attention kernels tile over N_kv with the online softmax)."""
    aria = os.path.join(os.path.dirname(YAML), "..", "Aria", "CSTS_Aria_Gaze_Forecast.yaml")
    g = _load("model_T32_B1_aria.npz")
    b = dev_batch(1, 32, 1003)
    for compute, tol in (("fp32", 1e-3), ("bf16", 5e-2)):
        cfg = load_yaml(aria, ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", 32, "CSTS_AMD.COMPUTE", compute])
        _MODELS.clear()
        m = build_model(cfg)
        m.load_state_dict(O.seeded_params(32, 256), strict=True)
        m.eval()
        with torch.no_grad():
            lg, v, a = m([b["video"]], b["audio"], return_embed=True)
        assert lg.shape == (1, 1, 32, 64, 64)
        assert rel_l2(lg, g["logits"].astype("float32")) < tol            # golden stored as fp16 (5e-4 of its own)
        if compute == "fp32":
            assert rel_l2(lg.flatten()[:4096], g["logits_head"]) < 1e-4   # fp32 slice of the golden
            assert rel_l2(v, g["v_emb"]) < 1e-4 and rel_l2(a, g["a_emb"]) < 1e-4
            heat = ops.frame_softmax(lg, 2.0)
            assert (heat.reshape(32, -1).argmax(-1).cpu().numpy() == g["argmax"]).all()
        del m
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ train-step parity
def _cos(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))


def _train_pass(m, cfg, batch):
    for p in m.parameters():
        p.grad = None
    loss, kld, nce, preds = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
    loss.backward()
    torch.cuda.synchronize()
    return loss, kld, nce, preds.detach()


def _check_train_fixture(m, g, loss, kld, nce, preds, *, loss_tol, norm_tol, slice_tol=None, cos_min=None, total_tol,
                         argmax_min=1.0, label="", abs_slice_tol=0.15):
    B, Tn = preds.shape[0], preds.shape[2]
    ref_loss, ref_kld, ref_nce = float(g["loss"]), float(g["kld"]), float(g["nce"])
    agree = float((preds.reshape(B, Tn, -1).argmax(-1).cpu().numpy() == g["argmax"]).mean())
    named = dict(m.named_parameters())
    worst_norm, worst_cos, worst_slice = 0.0, 1.0, 0.0
    bad = []
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if n == "classifier.bias":           # softmax is shift-invariant: the true gradient is 0 (rounding noise only)
            continue
        gr = named[n].grad
        assert gr is not None, n
        if ref_norm == 0.0:                  # EgoNCE branch at B = 1: identically zero in the reference too
            assert float(gr.double().norm()) < 1e-6, n
            continue
        err = abs(float(gr.double().norm()) - ref_norm) / ref_norm
        worst_norm = max(worst_norm, err)
        ref_slice = g[n.replace(".", "_") + "_g"]
        sl = gr.flatten()[:ref_slice.size]
        e, c = rel_l2(sl, ref_slice), _cos(sl, ref_slice)
        # error of the slice's elements relative to the TYPICAL element of this gradient tensor, and how much signal the
        # slice carries: 64 elements far below the tensor's rms are dominated by rounding noise in bf16 mode
        rms_t = ref_norm / gr.numel() ** 0.5
        rms_s = float(np.linalg.norm(ref_slice.astype(np.float64))) / ref_slice.size ** 0.5
        e_abs = float((sl.double().cpu() - torch.from_numpy(ref_slice.astype(np.float64))).norm()) / (ref_slice.size ** 0.5 * max(rms_t, rms_s))
        informative = rms_s >= 0.5 * rms_t
        worst_slice = max(worst_slice, e)
        if informative:
            worst_cos = min(worst_cos, c)
        fail = err > norm_tol or (slice_tol is not None and e >= slice_tol)
        if cos_min is not None:
            fail = fail or e_abs >= abs_slice_tol or (informative and c < cos_min)
        if fail:
            bad.append(f"{n}: |g| {float(gr.double().norm()):.4e} vs {ref_norm:.4e} (err {err:.2e}), slice rel-L2 {e:.2e}, cosine {c:.5f}, "
                       f"slice error / max(slice rms, tensor rms) {e_abs:.3f}, slice rms / tensor rms {rms_s / rms_t:.3f}")
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    ref_total = float(g["grad_total_norm"])
    print(f"\n[{label}] loss {float(loss):.6f} (ref {ref_loss:.6f})  kld {float(kld):.6f}/{ref_kld:.6f}  nce {float(nce):.5f}/{ref_nce:.5f}  "
          f"argmax agreement {agree:.3f}  worst grad-norm err {worst_norm:.2e}  worst slice rel-L2 {worst_slice:.2e}  "
          f"worst slice cosine {worst_cos:.5f}  total-norm err {abs(total - ref_total) / ref_total:.2e}")
    if bad:
        print(f"[{label}] gradients outside the bars:\n  " + "\n  ".join(bad))
    assert not bad, (label, len(bad), bad[:3])
    assert abs(float(loss) - ref_loss) <= loss_tol * abs(ref_loss)
    assert abs(float(kld) - ref_kld) <= loss_tol * max(abs(ref_kld), 1e-3)
    assert abs(float(nce) - ref_nce) <= max(10 * loss_tol * abs(ref_nce), 1e-3)
    assert abs(total - ref_total) <= total_tol * ref_total
    assert agree >= argmax_min, (label, agree)


def test_train_T16_fp32_vs_reference_golden():
    """The BENCHMARKED token grid (16x256^2), fp32 mode, B=2: loss / KLDiv / EgoNCE / 27 gradient norms + slices / total
    norm against the fixture generated by the imported reference (tools/train_avgaze_net.py:70-99)."""
    m, cfg = make_model("fp32", 16)
    g = _load("model_T16_B2_train.npz")
    loss, kld, nce, preds = _train_pass(m, cfg, dev_batch(2, 16, 1004))
    assert rel_l2(preds.flatten()[:4096], g["heat_head"]) < 1e-4
    _check_train_fixture(m, g, loss, kld, nce, preds, loss_tol=1e-4, norm_tol=2e-3, slice_tol=5e-3, total_tol=1e-3,
                         argmax_min=1.0, label="fp32 T16 B2")


@pytest.mark.parametrize("T_,fixture,seed", [(8, "model_T8_B2.npz", 1000), (16, "model_T16_B2_train.npz", 1004)])
def test_train_bf16_vs_reference_golden(T_, fixture, seed):
    """bf16 mode (the benchmarked one) at MODEL level, forward AND backward, against the reference fixture.  Stated bars
    (round 3: set at ~2 x what this build measures -- T8 / T16: loss error 2e-6 / 6e-6, worst gradient-norm error 8.5e-3 / 1.8e-2,
    total norm 2.2e-3 / 4.2e-3, worst informative slice cosine 0.9929 / 0.9917, argmax 1.0 / 31 of 32 frames):
    |loss - ref| / ref < 1e-3; per-tensor gradient-norm error < 3.6e-2; total gradient norm within 1e-2; per-frame argmax
    agreement >= 0.9375 (30 of 32 frames); on the stored 64-element gradient slices: rms error < 0.15 x max(slice rms, tensor
    rms), and cosine >= 0.983 where the slice carries signal (slice rms >= half the tensor rms).  Whole-tensor cosines of bf16-mode
    gradients are 0.98-0.998 (test_bf16_mode_gradients_all_tensors_vs_fp32_mode): bf16 operands with fp32 accumulation /
    residual stream / statistics / softmax / losses -- what torch.autocast gives the reference -- so a
    64-element sample sits around 0.99 and cannot be held to >= 0.99 tensor by tensor."""
    m, cfg = make_model("bf16", T_)
    g = _load(fixture)
    loss, kld, nce, preds = _train_pass(m, cfg, dev_batch(2, T_, seed))
    _check_train_fixture(m, g, loss, kld, nce, preds, loss_tol=1e-3, norm_tol=3.6e-2, cos_min=0.983, total_tol=1e-2,
                         argmax_min=0.9375, label=f"bf16 T{T_} B2")


@pytest.mark.parametrize("T_,amp_fixture,seed,B", [(8, "autocast_bf16_T8_B2.npz", 1000, 2), (16, "autocast_bf16_T16_B2.npz", 1004, 2)])
def test_bf16_mode_no_worse_than_reference_autocast(T_, amp_fixture, seed, B):
    """VERDICT round 3 item 2.  `autocast_bf16_*.npz` (oracle/gen_golden.py::gen_autocast) holds what the REFERENCE's own mixed
    precision costs on this path: the imported reference model, same seeded weights and batch, once in fp32 and once under
    torch.autocast(bfloat16) wrapped around forward + losses exactly as tools/train_avgaze_net.py:70-88 does.  Measured there:
    heat maps 3.4e-3 rel-L2 of its own fp32 run (logits 8.4e-3 ... 9.8e-3), arg-max 16/16 at T = 8 and 29/32 at T = 16, total
    gradient norm off by 0.5 % / 3.2 %.  This build's bf16 mode against the same fp32 outputs: 1.7e-3 / 4.7e-3, 16/16 and 31/32,
    0.2 % / 0.4 % -- half the reference's error.  Bar: every forward quantity and the gradient norms NO WORSE than the
    reference's autocast (x 1.0 for the forward, x 1.25 for the median gradient-norm error)."""
    a = _load(amp_fixture)
    m, cfg = make_model("bf16", T_)
    batch = dev_batch(B, T_, seed)
    loss, kld, nce, preds = _train_pass(m, cfg, batch)
    with torch.no_grad():
        logits, v, e = m([batch["video"]], batch["audio"], return_embed=True)
    heat_ref, logits_ref = torch.from_numpy(a["heat_ref"]), torch.from_numpy(a["logits_ref"])
    e_heat, e_logits = rel_l2(preds, heat_ref), rel_l2(logits, logits_ref)
    am = preds.reshape(B, T_, -1).argmax(-1).cpu().numpy()
    agree = float((am == a["argmax_ref"]).mean())
    e_loss = abs(float(loss) - float(a["loss_ref"])) / float(a["loss_ref"])
    ref_loss_err = abs(float(a["loss_amp"]) - float(a["loss_ref"])) / float(a["loss_ref"])
    named = dict(m.named_parameters())
    names = [str(n) for n in a["grad_names"]]
    keep = [i for i, n in enumerate(names) if n != "classifier.bias" and a["grad_norm_ref"][i] > 0]     # shift-invariant softmax: true gradient 0
    ours = np.array([abs(float(named[names[i]].grad.double().norm()) / a["grad_norm_ref"][i] - 1.0) for i in keep])
    theirs = np.abs(a["grad_norm_amp"][keep] / a["grad_norm_ref"][keep] - 1.0)
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
    e_total = abs(total / float(a["grad_total_norm_ref"]) - 1.0)
    ref_total = abs(float(a["grad_total_norm_amp"]) / float(a["grad_total_norm_ref"]) - 1.0)
    print(f"\n[bf16 mode vs the reference's bf16 autocast, T{T_} B{B}] heat rel-L2 {e_heat:.3e} (reference autocast {float(a['heat_rel_l2']):.3e})  "
          f"logits {e_logits:.3e} ({float(a['logits_rel_l2']):.3e})  argmax {agree:.4f} ({float(a['argmax_agree']):.4f})  "
          f"loss err {e_loss:.2e} ({ref_loss_err:.2e})  grad-norm err median {np.median(ours):.2e} ({np.median(theirs):.2e}) "
          f"max {ours.max():.2e} ({theirs.max():.2e})  total norm {e_total:.2e} ({ref_total:.2e})")
    assert e_heat <= float(a["heat_rel_l2"]) and e_logits <= float(a["logits_rel_l2"])
    assert agree >= float(a["argmax_agree"])
    assert e_loss <= max(ref_loss_err, 1e-5)
    assert np.median(ours) <= 1.25 * np.median(theirs) and ours.max() <= 1.25 * theirs.max()
    assert e_total <= max(ref_total, 5e-3)
    # north-star bar for reference: 1e-3 on the heat maps is an fp32-mode bar (met at 4e-7); bf16 operands cannot reach it --
    # the reference's own autocast is at 3.4e-3 -- so the bf16 mode is held to 2e-3 absolute here (measured 1.7e-3)
    assert e_heat < 2e-3


def test_bf16_mode_gradients_all_tensors_vs_fp32_mode():
    """EVERY gradient tensor of the bf16 mode against the fp32 mode of the same HIP model (which is pinned to the reference
    within 2e-3 by test_full_model_fp32_vs_reference_golden): cosine >= 0.97 and norm within 5 % for all 524 tensors
    (tensors whose true gradient is zero by softmax shift-invariance are skipped).  This is the test that catches a wrong
    bf16-only backward path anywhere in the model, not only in the 28 sampled tensors."""
    batch = dev_batch(2, 8, 1000)
    m, cfg = make_model("fp32")
    _train_pass(m, cfg, batch)
    ref = {n: p.grad.double().flatten().cpu() for n, p in m.named_parameters()}
    total = float(torch.sqrt(sum((v ** 2).sum() for v in ref.values())))
    m, cfg = make_model("bf16")
    _train_pass(m, cfg, batch)
    bad, worst = [], (1.0, "")
    for n, p in m.named_parameters():
        r, gq = ref[n], p.grad.double().flatten().cpu()
        if float(r.norm()) < 1e-6 * total:
            continue
        c, ratio = _cos(gq, r), float(gq.norm() / r.norm())
        worst = min(worst, (c, n))
        if c < 0.97 or not (0.95 < ratio < 1.05):
            bad.append(f"{n}: cosine {c:.5f}, |bf16| / |fp32| {ratio:.4f}")
    print(f"\n[bf16 vs fp32 mode, all tensors] worst cosine {worst[0]:.5f} ({worst[1]})")
    assert not bad, bad[:10]


def test_train_T32_aria_vs_reference_golden():
    """BASELINE config 5 geometry as a TRAIN step on one GPU: CSTS_Aria_Gaze_Forecast.yaml + DATA.NUM_FRAMES 32, B=1
    (N_kv = 4096: multi-tile keys, query splits, the dK/dV split reduction); fp32 gradients against the reference
    fixture, then the same step in bf16 mode with its stated bars; peak memory printed."""
    aria = os.path.join(os.path.dirname(YAML), "..", "Aria", "CSTS_Aria_Gaze_Forecast.yaml")
    g = _load("model_T32_B1_aria_train.npz")
    b = dev_batch(1, 32, 1003)
    for compute in ("fp32", "bf16"):
        cfg = load_yaml(aria, ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", 32, "CSTS_AMD.COMPUTE", compute])
        _MODELS.clear()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        m = build_model(cfg)
        m.load_state_dict(O.seeded_params(32, 256), strict=True)
        m.eval()
        loss, kld, nce, preds = _train_pass(m, cfg, b)
        print(f"\n[{compute} T32 B1] peak device memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
        if compute == "fp32":
            # slices of gradients reduced over up to 131 k tokens in fp32 (another summation order than the CPU's): 1e-2
            _check_train_fixture(m, g, loss, kld, nce, preds, loss_tol=1e-4, norm_tol=2e-3, slice_tol=1e-2, total_tol=1e-3,
                                 argmax_min=1.0, label="fp32 T32 B1 aria")
        else:
            # measured: loss error 5e-6, worst gradient-norm error 9.4e-3, total norm 2.6e-4, worst informative slice cosine
            # 0.9878, argmax 31 of 32 frames; bars at ~2 x that
            _check_train_fixture(m, g, loss, kld, nce, preds, loss_tol=1e-3, norm_tol=2e-2, cos_min=0.975, total_tol=2e-3,
                                 argmax_min=0.9375, label="bf16 T32 B1 aria")
        del m
    torch.cuda.empty_cache()


def test_spatial_audio_attn_train_gradient_fp32():
    """MVIT.SPATIAL_AUDIO_ATTN in TRAIN use: the gradient through the min-max-rescaled audio->pixel attention map
    (av_attention.py:356-370 -> custom_multimodal_builder.py:438-440) against the reference fixture."""
    m, cfg = make_model("fp32", 8, ("MVIT.SPATIAL_AUDIO_ATTN", True))
    g = _load("model_T8_B2_saa_train.npz")
    loss, kld, nce, preds = _train_pass(m, cfg, dev_batch(2, 8, 1005))
    _check_train_fixture(m, g, loss, kld, nce, preds, loss_tol=1e-4, norm_tol=2e-3, slice_tol=5e-3, total_tol=1e-3,
                         argmax_min=1.0, label="fp32 T8 B2 spatial-audio-attn")


def test_gradient_accumulation_and_failed_backward():
    """Deferred end-of-backward gradients (grouped weight gradients, LayerNorm / stencil second stages): (a) a second
    backward ACCUMULATES into existing .grad (2 x the single-pass gradient); (b) after a backward pass that raises, the
    next pass still produces complete, correct gradients (the dead pass's queues are discarded)."""
    m, cfg = make_model("bf16")
    batch = T.synthetic_batch(2, 8, 256, 31, DEV)
    old_mode, ops.GROUP_WGRADS = ops.GROUP_WGRADS, "always"
    try:
        assert ops.STENCIL_WGRAD_GROUPED
        ops.STENCIL_WGRAD_GROUPED = False     # (c) the pools' stencil weight gradients: one launch each ...
        try:
            _train_pass(m, cfg, batch)
            one_by_one = {n: p.grad.clone() for n, p in m.named_parameters() if "pool" in n and p.dim() == 5 and p.shape[1] == 1}
        finally:
            ops.STENCIL_WGRAD_GROUPED = True
        assert len(one_by_one) >= 40
        _train_pass(m, cfg, batch)            # ... == all of them in ONE grouped launch (longer token chunks: fp32 summation order)
        ref = {n: p.grad.clone() for n, p in m.named_parameters()}
        for n, g1 in one_by_one.items():
            assert rel_l2(ref[n], g1) < 1e-5, n
        loss, *_ = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
        loss.backward()                       # no zero_grad in between: accumulate
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            assert torch.allclose(p.grad, 2 * ref[n], rtol=1e-5, atol=1e-8), n

        class Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError("boom")

        for p in m.parameters():
            p.grad = None
        orig_pe = ops.patch_embed             # raise LATE in backward (just before the patch-embed nodes): by then the
        ops.patch_embed = lambda *a_, **k_: Boom.apply(orig_pe(*a_, **k_))   # queues hold the whole trunk's deferred work
        try:
            bad, *_ = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
        finally:
            ops.patch_embed = orig_pe
        with pytest.raises(RuntimeError, match="boom"):
            bad.backward()
        torch.cuda.synchronize()
        _train_pass(m, cfg, batch)
        for n, p in m.named_parameters():
            assert p.grad is not None and torch.equal(p.grad, ref[n]), n
    finally:
        ops.GROUP_WGRADS = old_mode
        ops.reset_deferred()


def test_train_step_runs_bf16_b4():
    """BASELINE config 3 shape: bs=4 train step (fwd + KLDiv + 0.05 EgoNCE + bwd + clip + AdamW), bf16 mode."""
    m, cfg = make_model("bf16")
    cfg.CSTS_AMD.FACTORED_ADAMW = True                   # the default in the 16-bit modes
    m.train()
    opt = T.construct_optimizer(m, cfg)
    batch = T.synthetic_batch(4, 8, 256, 1234, DEV)
    l0, _, _ = T.train_step(cfg, m, batch, opt, lr=1e-4)
    l1, _, _ = T.train_step(cfg, m, batch, opt, lr=1e-4)
    m.eval()
    assert torch.isfinite(l0) and torch.isfinite(l1)
    # the three fusion-conv weights have no materialised gradient: FusedAdamW formed it on the fly from its rank-(B T') factors
    fac = {id(m.vision_pool.weight), id(m.audio_pool.weight), id(m.audio_pool2.weight)}
    assert {id(p) for p in opt.last_factored_params} == fac
    assert opt._factored is None          # eager steps drop the factors after use (a step() without a fresh backward must not reuse them)
    assert all((p.grad is None) if id(p) in fac else (p.grad is not None and bool(torch.isfinite(p.grad).all())) for p in m.parameters())
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    _MODELS.clear()     # weights were updated: do not reuse


def test_size_independent_properties_full_size():
    """Properties at full size that need no oracle: heatmaps are distributions, batch rows are independent
    (clip i of a batch of 2 == the same clip alone), determinism of two identical calls."""
    m, cfg = make_model("fp32")
    batch = dev_batch(2, 8, 1000)
    with torch.no_grad():
        l2 = m([batch["video"]], batch["audio"])
        l2b = m([batch["video"]], batch["audio"])
        l1 = m([batch["video"][1:]], batch["audio"][1:])
        heat = ops.frame_softmax(l2, 2.0)
    assert torch.equal(l2, l2b)
    assert rel_l2(l1, l2[1:]) < 1e-5
    assert torch.allclose(heat.sum(dim=(-1, -2)), torch.ones(2, 1, 8, device=DEV), atol=1e-4)


def test_backward_is_bitwise_reproducible():
    """Two identical train steps (bf16 mode, two HIP streams, deferred reductions) give identical gradients: every
    cross-workgroup reduction in the library has a fixed order and no kernel races with another stream."""
    m, cfg = make_model("bf16")
    batch = T.synthetic_batch(2, 8, 256, 77, DEV)
    runs = []
    old_mode, ops.GROUP_WGRADS = ops.GROUP_WGRADS, "always"      # the grouped end-of-backward tail included
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        T.train_step(cfg, m, batch)
        torch.cuda.synchronize()
        runs.append({n: p.grad.clone() for n, p in m.named_parameters()})
    ops.GROUP_WGRADS = old_mode
    bad = [n for n in runs[0] if not torch.equal(runs[0][n], runs[1][n])]
    assert not bad, bad[:10]


def test_early_wgrad8_flush_is_bitwise_identical(monkeypatch):
    """CSTS_WGRAD8_EARLY_WGS: the 192 x 384-class weight gradients launched early, on few workgroups (each walking several items), on
    the side stream beside the rest of backward give bit-identical gradients: an item's arithmetic does not depend on the grid it is
    launched in, and the final callback joins the side stream before the gradients are handed to the parameters."""
    m, cfg = make_model("bf16")
    batch = T.synthetic_batch(2, 8, 256, 77, DEV)
    monkeypatch.setattr(ops, "GROUP_WGRADS", "always")

    def grads():
        for p in m.parameters():
            p.grad = None
        T.train_step(cfg, m, batch)
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters()}

    g0 = grads()
    monkeypatch.setattr(ops, "W8_EARLY_WGS", 64)
    g1 = grads()           # the first pass with the switch on only counts the class's problems
    assert ops._w8_total[0] > 0
    g2 = grads()           # this one launches them early
    bad = [n for n in g0 if not (torch.equal(g0[n], g1[n]) and torch.equal(g0[n], g2[n]))]
    assert not bad, bad[:10]


def test_rccl_single_rank_collective_paths():
    """The N>1 code path (bucketed async all-reduce from autograd hooks, EgoNCE all-gather with grad, fused scalar
    all-reduce) executed for real on RCCL with a 1-rank group: API / stream / thread semantics, not transport."""
    import torch.distributed as dist
    from csts_amd import distributed as du
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    du._FORCE = True
    try:
        m, cfg = make_model("bf16")
        for p in m.parameters():
            p.grad = None
        batch = T.synthetic_batch(2, 8, 256, 77, DEV)
        ref_loss, *_ = T.train_step(cfg, m, batch)           # plain single-process step
        ref = {n: p.grad.clone() for n, p in m.named_parameters()}
        wrapped = du.GradAllReduce(m, bucket_mb=64)
        assert len(wrapped._buckets) > 4
        for it in range(2):                                   # 2nd iteration uses the measured ready-order buckets
            loss, *_ = T.train_step(cfg, wrapped, batch)
            assert abs(float(loss) - float(ref_loss)) < 1e-5
            for n, p in m.named_parameters():
                assert torch.allclose(p.grad, ref[n], rtol=1e-4, atol=1e-7), n
        assert wrapped._observed and not wrapped._pending
        r = du.all_reduce([loss, loss * 2])
        assert abs(float(r[1]) - 2 * float(loss)) < 1e-6
    finally:
        du._FORCE = False
        ops.GROUP_WGRADS, ops.DEFER_REDUCTIONS = "capture", True     # GradAllReduce switches the grouped tail off process-wide
        dist.destroy_process_group()
        _MODELS.clear()


def test_graphed_train_step_matches_eager():
    """HIP-graph replay of the whole iteration == eager execution (same weights, same batch, eval-mode drop-path off)."""
    import copy
    m, cfg = make_model("bf16")
    m2 = copy.deepcopy(m)
    batch = T.synthetic_batch(2, 8, 256, 99, DEV)
    opt_e = T.construct_optimizer(m, cfg)
    opt_g = T.construct_optimizer(m2, cfg, capturable=True)
    state0 = copy.deepcopy(m2.state_dict())
    g = T.GraphedTrainStep(cfg, m2, opt_g, batch, warmup=1)
    m2.load_state_dict(state0)                                # undo warm-up / capture updates
    opt_g.reset_state() if hasattr(opt_g, "reset_state") else opt_g.state.clear()
    le = [float(T.train_step(cfg, m, batch, opt_e, lr=1e-4)[0]) for _ in range(2)]
    lg = [float(g.run(batch, lr=1e-4)[0]) for _ in range(2)]
    assert abs(le[0] - lg[0]) < 1e-4 and abs(le[1] - lg[1]) < 5e-3, (le, lg)
    w_e, w_g = m.blocks[5].mlp.fc1.weight, m2.blocks[5].mlp.fc1.weight
    assert rel_l2(w_g, w_e) < 1e-3
    _MODELS.clear()


@pytest.mark.parametrize("trunk_cut", [3, 0])
def test_segmented_train_step_matches_eager(trunk_cut):
    """The chain of HIP graphs (forward | eager losses | backward head | backward trunks, in two parts behind a second
    autograd cut in front of video block 3 or in one | clip + AdamW: the data-parallel step, here with one process and no
    collectives) == eager execution: same losses over two steps, same weights after."""
    import copy
    m, cfg = make_model("bf16")
    cfg.CSTS_AMD.TRUNK_CUT = trunk_cut
    m2 = copy.deepcopy(m)
    batch = T.synthetic_batch(2, 8, 256, 99, DEV)
    opt_e = T.construct_optimizer(m, cfg)
    opt_g = T.construct_optimizer(m2, cfg, capturable=True)
    state0 = copy.deepcopy(m2.state_dict())
    g = T.SegmentedTrainStep(cfg, m2, opt_g, batch, warmup=1)
    assert g.trunk_cut == trunk_cut and ("bwd_trunk_early" in g.graphs) == bool(trunk_cut)
    if trunk_cut:
        n_early = sum(p.numel() for p in g.early_params)
        assert 1.0e6 < n_early < 2.0e6 and len(g.early_params) + len(g.trunk_params) + len(g.head_params) == len(list(m2.parameters()))
    m2.load_state_dict(state0)                                # undo warm-up / capture updates
    opt_g.reset_state()
    le = [float(T.train_step(cfg, m, batch, opt_e, lr=1e-4)[0]) for _ in range(2)]
    lg = [float(g.run(batch, lr=1e-4, timed=True)[0]) for _ in range(2)]
    ms = g.segment_ms()
    print(f"\n[segmented step, T8 B2] eager losses {le}, graph-chain losses {lg}, segments (fwd, loss, bwd head, bwd trunk, opt) ms {[round(x, 3) for x in ms]}")
    assert abs(le[0] - lg[0]) < 1e-4 and abs(le[1] - lg[1]) < 5e-3, (le, lg)
    for name in ("blocks.5.mlp.fc1.weight", "blocks_audio.2.attn.pool_k.weight", "vision_pool.weight", "decode_block3.norm1.weight",
                 "pos_embed_spatial", "classifier.weight", "blocks.1.mlp.fc1.weight", "blocks.2.attn.qkv.weight",
                 "patch_embed.proj.weight", "blocks.3.norm1.weight"):
        w_e, w_g = dict(m.named_parameters())[name], dict(m2.named_parameters())[name]
        assert rel_l2(w_g, w_e) < 1e-3, name
    _MODELS.clear()
    ops.reset_deferred()


def test_segmented_step_with_rccl_one_rank():
    """The same chain with the collectives live (EgoNCE all-gather with grad, two flat-bucket all-reduces between the
    graphs) on a 1-rank RCCL group: gradients the optimizer sees == the plain step's, p.grad are views of the flat buckets."""
    import torch.distributed as dist
    from csts_amd import distributed as du
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29673")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    du._FORCE = True
    try:
        m, cfg = make_model("bf16")
        cfg.CSTS_AMD.FACTORED_ADAMW = False                  # this test reads EVERY gradient out of the buckets: keep the fusion convs' dW materialised
        batch = T.synthetic_batch(2, 8, 256, 77, DEV)
        for p in m.parameters():
            p.grad = None
        ref_loss, *_ = T.train_step(cfg, m, batch)           # plain single-process step, no optimizer
        ref = {n: p.grad.clone() for n, p in m.named_parameters()}
        wrapped = du.GradAllReduce(m, bucket_mb=64)
        opt = T.construct_optimizer(wrapped, cfg, capturable=True)
        opt.max_grad_norm = 0.0
        state0 = {k: v.clone() for k, v in m.state_dict().items()}
        g = T.SegmentedTrainStep(cfg, wrapped, opt, batch, warmup=1)
        assert not wrapped.hooks_enabled and ops.GROUP_WGRADS == "capture"
        m.load_state_dict(state0)
        opt.reset_state()
        loss, *_ = g.run(batch, lr=0.0)                      # lr 0: weights stay, gradients can be compared
        torch.cuda.synchronize()
        assert abs(float(loss) - float(ref_loss)) < 1e-4
        assert len(g.flat) == 3 and g.trunk_cut == 3          # head | late trunks | early video trunk (CSTS_AMD.TRUNK_CUT)
        flat_ptrs = [(f.data_ptr(), f.data_ptr() + f.numel() * 4) for f, _ in g.flat]
        for n, p in m.named_parameters():
            assert any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_ptrs), n
            assert torch.allclose(p.grad, ref[n], rtol=2e-3, atol=1e-6), n
    finally:
        du._FORCE = False
        ops.GROUP_WGRADS, ops.DEFER_REDUCTIONS = "capture", True
        ops.reset_deferred()
        dist.destroy_process_group()
        _MODELS.clear()


def test_crop_224_extension_properties():
    """16x224^2 (BASELINE.json's literal grid) as a labelled EXTENSION: CSTS_AMD.FUSION_KERNEL_FROM_GRID gives (1,7,7)
    fusion kernels.  The reference cannot run it, so there is no fixture: size-independent properties only --
    shapes, heat maps are distributions, clips of a batch are independent, two identical calls agree bit for bit, a train
    step yields finite gradients for every parameter."""
    cfg = load_yaml(YAML, ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", 16, "DATA.TRAIN_CROP_SIZE", 224,
                           "DATA.TEST_CROP_SIZE", 224, "CSTS_AMD.FUSION_KERNEL_FROM_GRID", True, "CSTS_AMD.COMPUTE", "fp32"])
    _MODELS.clear()
    torch.manual_seed(0)
    m = build_model(cfg).eval()
    assert tuple(m.vision_pool.weight.shape) == (768, 768, 1, 7, 7)
    batch = T.synthetic_batch(2, 16, 224, 5, DEV)
    assert batch["audio"].shape == (2, 1, 16, 224, 224) and batch["labels_hm"].shape == (2, 16, 56, 56)
    with torch.no_grad():
        l2 = m([batch["video"]], batch["audio"])
        l2b = m([batch["video"]], batch["audio"])
        l1 = m([batch["video"][1:]], batch["audio"][1:])
        heat = ops.frame_softmax(l2, 2.0)
    assert l2.shape == (2, 1, 16, 56, 56) and torch.equal(l2, l2b) and rel_l2(l1, l2[1:]) < 1e-5
    assert torch.allclose(heat.sum(dim=(-1, -2)), torch.ones(2, 1, 16, device=DEV), atol=1e-4)
    loss, kld, nce, _ = T.compute_loss(cfg, m, batch["video"], batch["audio"], batch["labels_hm"])
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    del m
    torch.cuda.empty_cache()
