"""CPU oracle for the CSTS hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 restatement of the reference algorithm for
the CSTS forward / loss path.  It is the *checker* for the HIP implementation
in ``csts_amd/``: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path never
routes through it.

Parity status: PINNED by fixtures generated from the imported reference
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``); the reference ships no
tests / golden vectors of its own, so those fixtures are
the only pin.  ``tests/test_oracle_golden.py`` checks this file against them.

Everything is written functionally over a flat ``{name: tensor}`` parameter
dict that uses the reference's ``state_dict`` names, so a reference checkpoint
drives it unchanged.  Citations are ``file:line`` under ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# geometry (slowfast/models/custom_multimodal_builder.py:115-216,270-299)
# --------------------------------------------------------------------------
def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:8-21."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


class BlockSpec:
    """Static geometry of one attention block."""

    def __init__(self, prefix, kind, dim, dim_out, heads, stride_q, stride_kv,
                 has_pool_q, has_pool_kv, mlp_hidden, drop_path=0.0):
        self.prefix = prefix          # state_dict prefix, e.g. "blocks.3"
        self.kind = kind              # "enc" | "dec" | "spatial" | "temporal"
        self.dim = dim
        self.dim_out = dim_out
        self.heads = heads
        self.stride_q = tuple(stride_q)
        self.stride_kv = tuple(stride_kv)
        self.has_pool_q = has_pool_q
        self.has_pool_kv = has_pool_kv
        self.mlp_hidden = mlp_hidden
        self.drop_path = drop_path

    def __repr__(self):
        return (f"BlockSpec({self.prefix}, {self.kind}, {self.dim}->{self.dim_out}, h={self.heads}, "
                f"sq={self.stride_q}, skv={self.stride_kv})")


def derive_geometry(embed_dim=96, num_heads=1, depth=16, mlp_ratio=4.0,
                    dim_mul=((1, 2.0), (3, 2.0), (14, 2.0)),
                    head_mul=((1, 2.0), (3, 2.0), (14, 2.0)),
                    pool_q_stride=((1, 1, 2, 2), (3, 1, 2, 2), (14, 1, 2, 2)),
                    pool_kv_stride_adaptive=(1, 8, 8),
                    drop_path_rate=0.2):
    """Restates the stride / width derivation of CSTS.__init__
    (custom_multimodal_builder.py:115-180 video, :184-216 audio, :271-299
    decoder, :232-268 fusion)."""
    dm = [1.0] * (depth + 1)
    hm = [1.0] * (depth + 1)
    for i, m in dim_mul:
        dm[int(i)] = m
    for i, m in head_mul:
        hm[int(i)] = m
    stride_q = [[] for _ in range(depth)]
    for row in pool_q_stride:
        stride_q[int(row[0])] = list(row[1:])
    # adaptive kv stride (:136-142)
    skv = list(pool_kv_stride_adaptive)
    stride_kv = []
    for i in range(depth):
        if len(stride_q[i]) > 0:
            skv = [max(skv[d] // stride_q[i][d], 1) for d in range(3)]
        stride_kv.append(list(skv))
    dpr = torch.linspace(0, drop_path_rate, depth).tolist()

    video = []
    heads, dim = num_heads, embed_dim
    for i in range(depth):
        heads = round_width(heads, hm[i])
        dim = round_width(dim, dm[i], divisor=heads)
        dim_out = round_width(dim, dm[i + 1], divisor=round_width(heads, hm[i + 1]))
        video.append(BlockSpec(f"blocks.{i}", "enc", dim, dim_out, heads,
                               stride_q[i] if stride_q[i] else (1, 1, 1), stride_kv[i],
                               has_pool_q=len(stride_q[i]) > 0, has_pool_kv=True,
                               mlp_hidden=int(dim * mlp_ratio), drop_path=dpr[i]))
    a_dim = [96, 192, 384, 768]
    a_out = [192, 384, 768, 768]
    a_heads = [1, 2, 4, 8]
    a_sq = [None, (1, 2, 2), (1, 2, 2), (1, 2, 2)]
    a_skv = [(1, 8, 8), (1, 4, 4), (1, 2, 2), (1, 1, 1)]
    audio = [BlockSpec(f"blocks_audio.{i}", "enc", a_dim[i], a_out[i], a_heads[i],
                       a_sq[i] or (1, 1, 1), a_skv[i], has_pool_q=a_sq[i] is not None,
                       has_pool_kv=True, mlp_hidden=int(a_dim[i] * mlp_ratio))
             for i in range(4)]
    token_dim = video[-1].dim_out
    fheads = video[-1].heads  # `num_heads` after the loop (:235,254)
    temporal = BlockSpec("temporal_fusion", "temporal", token_dim, token_dim, fheads, (1, 1, 1), (1, 1, 1),
                         False, False, int(token_dim * mlp_ratio))
    spatial = BlockSpec("spatial_fusion", "spatial", token_dim, token_dim, fheads, (1, 1, 1), (1, 1, 1),
                        False, False, int(token_dim * mlp_ratio))
    d_in = [768, 768, 384, 192]
    d_out = [768, 384, 192, 96]
    d_heads = [8, 4, 4, 2]
    d_sq = [(1, 2, 2), (1, 2, 2), (1, 2, 2), (2, 1, 1)]
    d_skv = [(1, 2, 2), (1, 4, 4), (1, 8, 8), (1, 16, 16)]
    decoder = [BlockSpec(f"decode_block{i + 1}", "dec", d_in[i], d_out[i], d_heads[i], d_sq[i], d_skv[i],
                         True, True, int(d_out[i] * mlp_ratio))  # hidden = 4*dim_out (attention.py:444)
               for i in range(4)]
    return {"video": video, "audio": audio, "temporal": temporal, "spatial": spatial,
            "decoder": decoder, "token_dim": token_dim}


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def _fold(x, heads, thw):
    """(B, h, L, hd) -> (B*h, hd, T, H, W)   (attention.py:29-31)."""
    B, h, L, hd = x.shape
    T, H, W = thw
    return x.reshape(B * h, T, H, W, hd).permute(0, 4, 1, 2, 3).contiguous()


def _unfold(x, B, heads):
    """(B*h, hd, T, H, W) -> (B, h, L', hd), thw'   (attention.py:35-37)."""
    thw = [x.shape[2], x.shape[3], x.shape[4]]
    L = thw[0] * thw[1] * thw[2]
    hd = x.shape[1]
    return x.reshape(B, heads, hd, L).transpose(2, 3), thw


def pool_conv_ln(x, thw, weight, stride, ln_w, ln_b):
    """attention_pool with a depthwise Conv3d k=3 p=1 + LayerNorm(hd, eps=1e-5)
    (attention.py:11-49 with the pools of :104-116)."""
    B, h, L, hd = x.shape
    y = F.conv3d(_fold(x, h, thw), weight, None, stride=tuple(stride), padding=1, groups=hd)
    y, thw2 = _unfold(y, B, h)
    y = F.layer_norm(y, (hd,), ln_w, ln_b, 1e-5)
    return y, thw2


def upsample_conv_ln(x, thw, weight, stride, ln_w, ln_b):
    """attention_upsample with depthwise ConvTranspose3d k=3 p=1 output_padding=s-1
    + LayerNorm(hd, eps=1e-5)   (attention.py:251-289, :323,344-349)."""
    B, h, L, hd = x.shape
    outpad = tuple(0 if s == 1 else s - 1 for s in stride)
    y = F.conv_transpose3d(_fold(x, h, thw), weight, None, stride=tuple(stride), padding=1,
                           output_padding=outpad, groups=hd)
    y, thw2 = _unfold(y, B, h)
    y = F.layer_norm(y, (hd,), ln_w, ln_b, 1e-5)
    return y, thw2


def maxpool_skip(x, thw, stride):
    """Residual-path MaxPool3d, kernel s+1 where s>1, padding k//2
    (attention.py:193-195,234-236,240)."""
    B, L, C = x.shape
    k = [s + 1 if s > 1 else s for s in stride]
    p = [kk // 2 for kk in k]
    y = F.max_pool3d(_fold(x.unsqueeze(1), 1, thw), k, tuple(stride), p)
    y, thw2 = _unfold(y, B, 1)
    return y.squeeze(1), thw2


def trilinear_skip(x, thw, stride):
    """Decoder residual path nn.Upsample(scale_factor=stride, mode='trilinear')
    (attention.py:463-467,471)."""
    B, L, C = x.shape
    y = F.interpolate(_fold(x.unsqueeze(1), 1, thw), scale_factor=tuple(float(s) for s in stride),
                      mode="trilinear")
    y, thw2 = _unfold(y, B, 1)
    return y.squeeze(1), thw2


def mlp(x, P, prefix):
    """Mlp.forward, exact-erf GELU (common.py:26-34)."""
    h = F.linear(x, P[prefix + ".fc1.weight"], P[prefix + ".fc1.bias"])
    h = F.gelu(h)
    return F.linear(h, P[prefix + ".fc2.weight"], P[prefix + ".fc2.bias"])


def spatial_mask(T, HW, device):
    """Same-frame block mask of SpatialAttention (av_attention.py:336-344):
    0 where query and key belong to the same frame, 1e8 elsewhere."""
    THW = T * HW
    N = THW + T
    frame = torch.cat([torch.arange(THW, device=device) // HW, torch.arange(T, device=device)])
    same = frame[:, None] == frame[None, :]
    off = torch.full((N, N), 1e8, device=device)
    off[same] = 0.0
    return off


def attention_core(q, k, v, scale, mask=None):
    """softmax(q k^T * scale [- mask]) v  (attention.py:154-158; av_attention.py:334-350)."""
    attn = (q @ k.transpose(-2, -1)) * scale
    if mask is not None:
        attn = attn - mask
    attn = attn.softmax(dim=-1)
    return attn @ v, attn


def _drop(x_branch, keep_mask, drop_prob):
    """drop_path with an explicit per-sample keep mask (common.py:46-59):
    x / keep_prob * mask.  keep_mask None == eval mode.  The reference draws
    TWO independent masks per block (attention branch, MLP branch:
    attention.py:242,247), hence the (mask_attn, mask_mlp) pairs below."""
    if keep_mask is None or drop_prob == 0.0:
        return x_branch
    keep = 1.0 - drop_prob
    return x_branch.div(keep) * keep_mask.view(-1, 1, 1).to(x_branch.dtype)


def block_forward(x, thw, P: Params, spec: BlockSpec, keep_masks=None, want_attn=False,
                  spatial_audio_attn=False):
    """MultiScaleBlock.forward (attention.py:238-248), MultiScaleDecoderBlock.forward
    (:469-479), SpatialBlock.forward (av_attention.py:451-473), TemporalBlock.forward (:233-250)."""
    p = spec.prefix
    km_attn, km_mlp = keep_masks if keep_masks is not None else (None, None)
    B, N, C = x.shape
    h = spec.heads
    hd = C // h
    scale = hd ** -0.5
    xn = F.layer_norm(x, (C,), P[p + ".norm1.weight"], P[p + ".norm1.bias"], 1e-6)
    qkv = F.linear(xn, P[p + ".attn.qkv.weight"], P[p + ".attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, h, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q_thw = list(thw)
    if spec.kind == "dec":
        q, q_thw = upsample_conv_ln(q, thw, P[p + ".attn.upsample_q.weight"], spec.stride_q,
                                    P[p + ".attn.norm_q.weight"], P[p + ".attn.norm_q.bias"])
    elif spec.has_pool_q:
        q, q_thw = pool_conv_ln(q, thw, P[p + ".attn.pool_q.weight"], spec.stride_q,
                                P[p + ".attn.norm_q.weight"], P[p + ".attn.norm_q.bias"])
    if spec.has_pool_kv:
        k, _ = pool_conv_ln(k, thw, P[p + ".attn.pool_k.weight"], spec.stride_kv,
                            P[p + ".attn.norm_k.weight"], P[p + ".attn.norm_k.bias"])
        v, _ = pool_conv_ln(v, thw, P[p + ".attn.pool_v.weight"], spec.stride_kv,
                            P[p + ".attn.norm_v.weight"], P[p + ".attn.norm_v.bias"])
    mask = None
    if spec.kind == "spatial":
        mask = spatial_mask(thw[0], thw[1] * thw[2], x.device)
    o, attn = attention_core(q, k, v, scale, mask)
    Nq = q.shape[2]
    o = o.transpose(1, 2).reshape(B, Nq, C)
    x_block = F.linear(o, P[p + ".attn.proj.weight"], P[p + ".attn.proj.bias"])

    if spec.kind == "enc" and spec.has_pool_q:
        x_res, _ = maxpool_skip(x, thw, spec.stride_q)
    elif spec.kind == "dec":
        x_res, _ = trilinear_skip(x, thw, spec.stride_q)
    else:
        x_res = x
    x = x_res + _drop(x_block, km_attn, spec.drop_path)
    x_norm = F.layer_norm(x, (C,), P[p + ".norm2.weight"], P[p + ".norm2.bias"], 1e-6)
    x_mlp = mlp(x_norm, P, p + ".mlp")
    if spec.dim != spec.dim_out:
        x = F.linear(x_norm, P[p + ".proj.weight"], P[p + ".proj.bias"])
    x = x + _drop(x_mlp, km_mlp, spec.drop_path)
    extra = None
    if spec.kind == "spatial" and spatial_audio_attn:
        # av_attention.py:360-370
        T, H, W = thw
        HW, THW = H * W, T * H * W
        aa = torch.stack([attn[:, :, THW + t, HW * t:HW * (t + 1)] for t in range(T)], dim=2)
        amax = aa.max(dim=-1, keepdim=True)[0]
        amin = aa.min(dim=-1, keepdim=True)[0]
        extra = ((aa - amin) / (amax - amin + 1e-8)).reshape(B, h, T, H, W)
    elif want_attn:
        extra = attn
    return x, q_thw, extra


def patch_embed(x, w, b, stride, padding):
    """PatchEmbed.forward: Conv3d + flatten(2).transpose(1, 2) (stem_helper.py:35-38)."""
    y = F.conv3d(x, w, b, stride=tuple(stride), padding=tuple(padding))
    return y.flatten(2).transpose(1, 2)


def fusion_conv(x_tok, thw, w, b):
    """Conv3d(C, C, (1,8,8)) over folded tokens -> (B, T, C)
    (custom_multimodal_builder.py:227-229,420-421,442-445)."""
    B, N, C = x_tok.shape
    y = F.conv3d(x_tok.reshape(B, *thw, C).permute(0, 4, 1, 2, 3), w, b)
    return y.squeeze(-1).squeeze(-1).permute(0, 2, 1)


# --------------------------------------------------------------------------
# full model (custom_multimodal_builder.py:343-498)
# --------------------------------------------------------------------------
def csts_forward(P: Params, video: torch.Tensor, audio: torch.Tensor, num_frames: int, crop: int,
                 return_embed=False, return_spatial_attn=False, return_temporal_attn=False,
                 spatial_audio_attn=False, keep_masks: Optional[Dict[str, torch.Tensor]] = None,
                 geometry=None, patch_stride=(2, 4, 4), patch_padding=(1, 3, 3), taps=None):
    """Forward of the CSTS model.  ``video`` (B,3,T,S,S), ``audio`` (B,1,T,S,S).
    ``keep_masks`` maps block prefix -> ((B,), (B,)) 0/1 keep masks (train-mode drop-path);
    None means eval mode.  ``taps`` (optional dict) receives intermediate tensors."""
    G = geometry or derive_geometry()
    km = keep_masks or {}
    T, H, W = num_frames // patch_stride[0], crop // patch_stride[1], crop // patch_stride[2]
    x = patch_embed(video, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], patch_stride, patch_padding)
    y = patch_embed(audio, P["patch_embed_audio.proj.weight"], P["patch_embed_audio.proj.bias"],
                    patch_stride, patch_padding)
    # separable positional embedding (:362-370)
    pos = P["pos_embed_spatial"].repeat(1, T, 1) + torch.repeat_interleave(P["pos_embed_temporal"], H * W, dim=1)
    pos_a = P["pos_embed_spatial_audio"].repeat(1, T, 1) + torch.repeat_interleave(
        P["pos_embed_temporal_audio"], H * W, dim=1)
    x = x + pos
    y = y + pos_a
    thw = [T, H, W]
    thw_a = [T, H, W]
    inter = [(x, thw)]
    vb, ab = G["video"], G["audio"]

    def run(xx, tt, specs):
        for s in specs:
            xx, tt, _ = block_forward(xx, tt, P, s, km.get(s.prefix))
        return xx, tt

    # interleaved stages (:387-411)
    x, thw = run(x, thw, vb[:1]); inter.append((x, thw)); y, thw_a = run(y, thw_a, ab[:1])
    x, thw = run(x, thw, vb[1:3]); inter.append((x, thw)); y, thw_a = run(y, thw_a, ab[1:2])
    x, thw = run(x, thw, vb[3:14]); inter.append((x, thw)); y, thw_a = run(y, thw_a, ab[2:3])
    x, thw = run(x, thw, vb[14:]); y, thw_a = run(y, thw_a, ab[3:])
    if taps is not None:
        taps["enc_video"] = x
        taps["enc_audio"] = y

    # spatial fusion (:415-432)
    B, Nv, C = x.shape
    y_sp = fusion_conv(y, thw_a, P["audio_pool.weight"], P["audio_pool.bias"])          # (B, T, C)
    av_sp = torch.cat([x, y_sp], dim=1)
    av_sp, _, sp_extra = block_forward(av_sp, thw, P, G["spatial"], want_attn=return_spatial_attn,
                                       spatial_audio_attn=spatial_audio_attn)
    x_spatial = av_sp[:, :Nv, :]
    # temporal fusion (:435-451)
    x_t = x.reshape(B, *thw, C)
    if spatial_audio_attn:
        x_t = x_t * sp_extra.mean(dim=1).unsqueeze(-1)
    x_tmp = fusion_conv(x_t.reshape(B, Nv, C), thw, P["vision_pool.weight"], P["vision_pool.bias"])
    y_tmp = fusion_conv(y, thw_a, P["audio_pool2.weight"], P["audio_pool2.bias"])
    av_t = torch.cat([x_tmp, y_tmp], dim=1)                                              # (B, 2T, C)
    av_t, _, t_extra = block_forward(av_t, (2, 2, 2), P, G["temporal"], want_attn=return_temporal_attn)
    # re-weight (:454-461)
    Tn = x_tmp.shape[1]
    x_w, y_w = av_t[:, :Tn, :], av_t[:, Tn:, :]
    x_rw = (x_spatial.reshape(B, *thw, C) * x_w[:, :, None, None, :]).reshape(B, Nv, C)
    y_rw = (y.reshape(B, *thw_a, C) * y_w[:, :, None, None, :]).reshape(B, y.shape[1], C)
    if taps is not None:
        taps["x_reweight"] = x_rw
        taps["y_reweight"] = y_rw

    # decoder (:466-481)
    dec = G["decoder"]
    feat, thw = x_rw, list(thw)
    for i, s in enumerate(dec):
        feat, thw, _ = block_forward(feat, thw, P, s, km.get(s.prefix))
        if i < 3:
            feat = feat + inter[-1 - i][0]
    feat = feat.reshape(B, *thw, feat.shape[2]).permute(0, 4, 1, 2, 3)
    en, en_thw = inter[0]
    en = en.reshape(B, *en_thw, en.shape[2]).permute(0, 4, 1, 2, 3)
    feat = feat + F.interpolate(en, size=(en_thw[0] * 2, en_thw[1], en_thw[2]), mode="trilinear")
    logits = F.conv3d(feat, P["classifier.weight"], P["classifier.bias"])               # (B,1,2T',H,W)

    if not return_embed and not return_spatial_attn and not return_temporal_attn:
        return logits
    if not return_embed:
        out = [logits]
        if return_spatial_attn:
            out.append(sp_extra)
        if return_temporal_attn:
            out.append(t_extra)
        return out
    v_emb = F.linear(x_rw.mean(dim=1), P["vision_proj.weight"], P["vision_proj.bias"])
    a_emb = F.linear(y_rw.mean(dim=1), P["audio_proj.weight"], P["audio_proj.bias"])
    return [logits, v_emb, a_emb]


# --------------------------------------------------------------------------
# losses (slowfast/utils/utils.py:5-24, slowfast/models/losses.py:51-82,152-170)
# --------------------------------------------------------------------------
def frame_softmax(logits, temperature=2.0):
    B, T, H, W = logits.shape[0], logits.shape[2], logits.shape[3], logits.shape[4]
    p = F.softmax(logits.reshape(B, -1, T, H * W) / temperature, dim=-1)
    return p.reshape(B, -1, T, H, W)


def kldiv(pred, target):
    B, T, H, W = pred.shape[0], pred.shape[2], pred.shape[3], pred.shape[4]
    p = pred.reshape(B, T, -1)
    logp = torch.log(p + 1e-10)
    logq = torch.log(target.reshape(B, T, -1) + 1e-10)
    kl = (p * logp).sum(-1) - (p * logq).sum(-1)
    return (kl.sum(-1) / (T * math.log(H * W))).mean()


def sim_matrix(a, b, eps=1e-8):
    an = a.norm(dim=1)[:, None]
    bn = b.norm(dim=1)[:, None]
    a_n = a / torch.max(an, eps * torch.ones_like(an))
    b_n = b / torch.max(bn, eps * torch.ones_like(bn))
    return a_n @ b_n.t()


def egonce(sim, temperature=0.05):
    """Symmetric InfoNCE over the diagonal (losses.py:157-170), without the
    hard-coded ``.cuda()`` of :158."""
    n = sim.shape[0]
    eye = torch.eye(n, device=sim.device) > 0
    i_sm = F.softmax(sim / temperature, dim=1)
    j_sm = F.softmax(sim.t() / temperature, dim=1)
    li = torch.log((i_sm * eye).sum(1)).sum() / n
    lj = torch.log((j_sm * eye).sum(1)).sum() / n
    return -li - lj


def csts_loss(logits, v_emb, a_emb, labels_hm, alpha=0.05):
    """The loss of train_avgaze_net.py:84-88 (LOSS_FUNC kldiv+egonce)."""
    kld = kldiv(frame_softmax(logits, 2.0), labels_hm)
    nce = egonce(sim_matrix(v_emb, a_emb))
    return kld + alpha * nce, kld, nce


# --------------------------------------------------------------------------
# deterministic parameter generator shared by fixtures, tests and the bench
# --------------------------------------------------------------------------
def param_manifest(num_frames=8, crop=256, with_nce=True, geometry=None) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list equal to the reference ``state_dict()``
    (checked against tests/golden/manifest_T*.json captured from the reference)."""
    G = geometry or derive_geometry()
    T, HW = num_frames // 2, (crop // 4) ** 2
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("pos_embed_spatial", (1, HW, 96)), ("pos_embed_temporal", (1, T, 96)),
        ("pos_embed_spatial_audio", (1, HW, 96)), ("pos_embed_temporal_audio", (1, T, 96)),
        ("patch_embed.proj.weight", (96, 3, 3, 7, 7)), ("patch_embed.proj.bias", (96,)),
        ("patch_embed_audio.proj.weight", (96, 1, 3, 7, 7)), ("patch_embed_audio.proj.bias", (96,)),
    ]

    def block(s: BlockSpec):
        p, C, hd = s.prefix, s.dim, s.dim // s.heads
        r = [(p + ".norm1.weight", (C,)), (p + ".norm1.bias", (C,)),
             (p + ".attn.qkv.weight", (3 * C, C)), (p + ".attn.qkv.bias", (3 * C,)),
             (p + ".attn.proj.weight", (C, C)), (p + ".attn.proj.bias", (C,))]
        if s.kind == "dec":
            r += [(p + ".attn.upsample_q.weight", (hd, 1, 3, 3, 3)),
                  (p + ".attn.norm_q.weight", (hd,)), (p + ".attn.norm_q.bias", (hd,))]
        elif s.has_pool_q:
            r += [(p + ".attn.pool_q.weight", (hd, 1, 3, 3, 3)),
                  (p + ".attn.norm_q.weight", (hd,)), (p + ".attn.norm_q.bias", (hd,))]
        if s.has_pool_kv:
            r += [(p + ".attn.pool_k.weight", (hd, 1, 3, 3, 3)),
                  (p + ".attn.norm_k.weight", (hd,)), (p + ".attn.norm_k.bias", (hd,)),
                  (p + ".attn.pool_v.weight", (hd, 1, 3, 3, 3)),
                  (p + ".attn.norm_v.weight", (hd,)), (p + ".attn.norm_v.bias", (hd,))]
        r += [(p + ".norm2.weight", (C,)), (p + ".norm2.bias", (C,)),
              (p + ".mlp.fc1.weight", (s.mlp_hidden, C)), (p + ".mlp.fc1.bias", (s.mlp_hidden,)),
              (p + ".mlp.fc2.weight", (s.dim_out, s.mlp_hidden)), (p + ".mlp.fc2.bias", (s.dim_out,))]
        if s.dim != s.dim_out:
            r += [(p + ".proj.weight", (s.dim_out, C)), (p + ".proj.bias", (s.dim_out,))]
        return r

    for s in G["video"]:
        out += block(s)
    for s in G["audio"]:
        out += block(s)
    D = G["token_dim"]
    if with_nce:
        out += [("vision_proj.weight", (256, D)), ("vision_proj.bias", (256,)),
                ("audio_proj.weight", (256, D)), ("audio_proj.bias", (256,))]
    for n in ("vision_pool", "audio_pool", "audio_pool2"):
        out += [(n + ".weight", (D, D, 1, 8, 8)), (n + ".bias", (D,))]
    out += block(G["temporal"])
    out += block(G["spatial"])
    for s in G["decoder"]:
        out += block(s)
    out += [("classifier.weight", (1, 96, 1, 1, 1)), ("classifier.bias", (1,))]
    return out


def seeded_tensor(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """Documented per-tensor generator: the same function initialises the
    reference model (in gen_golden.py), the oracle and the HIP model, so
    188 M weights never need to be committed.  Non-trivial biases / LN affine
    on purpose, so that every bias path is exercised."""
    import zlib
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if name.startswith("pos_embed"):
        return r * 0.1
    if len(shape) == 1:
        if ".norm" in name and name.endswith(".weight"):
            return 1.0 + 0.1 * r
        return 0.05 * r
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return r * (0.8 / math.sqrt(fan_in))


def seeded_params(num_frames=8, crop=256, with_nce=True, seed=0) -> Params:
    return {n: seeded_tensor(n, s, seed) for n, s in param_manifest(num_frames, crop, with_nce)}


def synthetic_batch(B: int, num_frames: int = 8, crop: int = 256, seed: int = 1000,
                    device="cpu") -> Dict[str, torch.Tensor]:
    """Synthetic clip batch following the dataset contract (SURVEY.md 8(d);
    ego4d_avgaze_forecast.py:214-221,294-335): normalised uint8 video,
    log-power STFT windows of 24 kHz noise, 19x19-Gaussian gaze heatmaps."""
    g = torch.Generator().manual_seed(seed)
    T, S = num_frames, crop
    u = torch.randint(0, 256, (B, 3, T, S, S), generator=g, dtype=torch.int16).float()
    video = (u / 255.0 - 0.45) / 0.225
    # 5 s of 24 kHz noise (+440 Hz tone) -> STFT(n_fft 511, hop 120, win 240) -> log power
    n = 24000 * 5
    wav = 0.1 * torch.randn(B, n, generator=g) + 0.05 * torch.sin(
        2 * math.pi * 440.0 * torch.arange(n) / 24000.0)[None]
    spec = torch.stft(wav, n_fft=511, hop_length=120, win_length=240, window=torch.hann_window(240),
                      center=True, pad_mode="constant", return_complex=True)
    logp = torch.log(spec.abs() ** 2 + 1e-6)                      # (B, 256, cols)
    cols = logp.shape[-1]
    audio = torch.empty(B, 1, T, S, S)
    for t in range(T):
        c = int(round((t + 0.5) / T * cols))
        c = max(128, min(cols - 129, c))
        audio[:, 0, t] = logp[:, :, c - 128:c + 128]
    # gaze heatmaps: 19x19 Gaussian (OpenCV sigma for ksize 19 = 3.2) at a random centre, sum 1
    hm = torch.zeros(B, T, 64, 64)
    yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(64.0), indexing="ij")
    cx = torch.rand(B, T, generator=g) * 63
    cy = torch.rand(B, T, generator=g) * 63
    for b in range(B):
        for t in range(T):
            d2 = (xx - cx[b, t].round()) ** 2 + (yy - cy[b, t].round()) ** 2
            k = torch.exp(-d2 / (2 * 3.2 ** 2)) * ((xx - cx[b, t].round()).abs() <= 9) * (
                (yy - cy[b, t].round()).abs() <= 9)
            hm[b, t] = k / k.sum()
    labels = torch.stack([cx / 63, cy / 63, torch.zeros_like(cx)], dim=-1).double()
    out = {"video": video, "audio": audio, "labels_hm": hm, "labels": labels}
    return {k: v.to(device) for k, v in out.items()}


# --------------------------------------------------------------------------------------- evaluation metric
def minmax_rescale(preds):
    """Per-frame min-max rescale applied before the metric (tools/test_avgaze_net.py:66-68, train_avgaze_net.py:125-127)."""
    flat = preds.reshape(preds.shape[:-2] + (preds.shape[-1] * preds.shape[-2],))
    mn, mx = flat.min(dim=-1, keepdim=True)[0], flat.max(dim=-1, keepdim=True)[0]
    return ((flat - mn) / (mx - mn + 1e-6)).reshape(preds.shape)


def adaptive_f1(preds, labels_hm, labels, dataset):
    """Best F1 over the dataset's threshold sweep on fixation frames (slowfast/utils/metrics.py:9-74), without the
    (n_thr, B, T, H, W) temporaries: per-frame counts per threshold, then the same means / f1 / argmax."""
    import numpy as np
    if "forecast" in dataset and "aria" not in dataset:
        thresholds = np.linspace(0.01, 0.07, 31)                      # metrics.py:35-37
    elif "forecast" in dataset and "aria" in dataset:
        thresholds = np.linspace(0.0, 0.02, 21)                       # :38-40
    else:
        thresholds = np.linspace(0, 0.02, 11)                         # :41-43
    p = preds.squeeze(1)
    lab = (labels_hm > 0.001)                                          # :47
    tp, fgp = [], []
    for t in thresholds:
        pr = p > float(t)                                              # :49
        tp.append((pr & lab).sum(dim=(2, 3)).float())
        fgp.append(pr.sum(dim=(2, 3)).float())
    tp, fgp = torch.stack(tp), torch.stack(fgp)                        # (n_thr, B, T)
    fgl = lab.sum(dim=(2, 3)).float().unsqueeze(0).expand_as(tp)
    fixation_idx = 1 if dataset == "egteagaze" else 0                  # :56-62
    tracked = torch.where(labels.reshape(-1, labels.shape[2])[:, 2] == fixation_idx)[0]     # :63-64
    tp, fgp, fgl = [x.reshape(x.shape[0], -1).index_select(1, tracked) for x in (tp, fgp, fgl)]
    recall = (tp / (fgl + 1e-6)).mean(dim=1)                           # :68
    precision = (tp / (fgp + 1e-6)).mean(dim=1)                        # :69
    f1 = (2 * recall * precision) / (recall + precision + 1e-6)        # :70
    i = int(torch.argmax(f1))
    return float(f1[i]), float(recall[i]), float(precision[i]), thresholds[i]


# --------------------------------------------------------------------------------------- input pipeline
# PARITY UNPINNED against the reference for this block: the reference computes these with librosa / OpenCV, which are
# absent from the build image, so the functions below restate the PUBLISHED algorithms (librosa.stft with center=True,
# pad_mode='constant', periodic Hann window padded to n_fft; cv2.getGaussianKernel's closed form) at the reference's call
# sites; the STFT is cross-checked against torch.stft in tests/test_oracle_golden.py.
def frames_normalize(frames_u8, mean=(0.45, 0.45, 0.45), std=(0.225, 0.225, 0.225)):
    """uint8 (T, H, W, C) -> float (C, T, H, W): slowfast/datasets/utils.py:290-307 (tensor_normalize) + the permute of
    ego4d_avgaze_forecast.py:296."""
    x = frames_u8.float() / 255.0
    x = (x - torch.tensor(mean)) / torch.tensor(std)
    return x.permute(3, 0, 1, 2).contiguous()


def stft_logpower(wav, n_fft=511, hop=120, win=240, eps=1e-6):
    """data/preprocess.py:287-290: log(|librosa.stft(y, n_fft=511, window='hann', hop_length=120, win_length=240,
    pad_mode='constant')|^2 + 1e-6) for 24 kHz audio (10 ms window, 5 ms step) -> (n_fft//2 + 1, 1 + len(y)//hop)."""
    import numpy as np
    y = np.asarray(wav, dtype=np.float64)
    k = np.arange(win)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / win)                      # scipy get_window('hann', win, fftbins=True)
    lpad = (n_fft - win) // 2
    wfull = np.zeros(n_fft)
    wfull[lpad:lpad + win] = w                                          # librosa.util.pad_center
    ypad = np.pad(y, n_fft // 2, mode="constant")                       # center=True
    nfr = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(nfr)[:, None]
    S = np.fft.rfft(ypad[idx] * wfull[None, :], n=n_fft, axis=1).T      # (bins, frames)
    return np.log((S * np.conj(S)).real + eps).astype(np.float32)


def audio_windows(spec, frames_idx, frame_length):
    """ego4d_avgaze_forecast.py:214-219: T windows of 256 spectrogram columns centred on the sampled video frames."""
    import numpy as np
    cols = spec.shape[1]
    idx = np.round(np.asarray(frames_idx, dtype=np.float64) / frame_length * cols).astype(np.int64)   # torch.round: half to even
    idx = np.clip(idx, 128, cols - 1 - 128)
    return np.stack([spec[:, i - 128:i + 128] for i in idx], axis=0)[None], idx


def gaze_heatmaps(labels_xy, T, H=64, W=64, ksize=19):
    """ego4d_avgaze_forecast.py:318-326 + _get_gaussian_map (:404-422): a ksize x ksize OpenCV Gaussian
    (sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8) pasted at round(x*W), round(y*H), clipped at the border, renormalised to sum 1
    (uniform 1/(H*W) when the gaze falls outside)."""
    import numpy as np
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    i = np.arange(ksize) - (ksize - 1) / 2
    k1 = np.exp(-(i * i) / (2 * sigma * sigma))
    k1 = (k1 / k1.sum()).astype(np.float32)                              # ktype CV_32F
    k2 = np.outer(k1, k1)
    out = np.zeros((T, H, W), dtype=np.float64)
    r = (ksize - 1) // 2
    for t in range(T):
        mu_x, mu_y = round(float(labels_xy[t][0]) * W), round(float(labels_xy[t][1]) * H)   # Python round: half to even
        left, right = max(mu_x - r, 0), min(mu_x + r, W - 1)
        top, bottom = max(mu_y - r, 0), min(mu_y + r, H - 1)
        if not (left >= right or top >= bottom):
            out[t, top:bottom + 1, left:right + 1] = k2[r - mu_y + top:r + bottom - mu_y + 1, r - mu_x + left:r + right - mu_x + 1]
        s = out[t].sum()
        if s == 0:
            out[t] += 1.0 / (H * W)
        elif s != 1:
            out[t] /= s
    return out.astype(np.float32)
