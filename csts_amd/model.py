"""CSTS model: reference-identical nn.Module surface, MI355X-native forward/backward.

Drop-in for ``slowfast.models.custom_multimodal_builder.CSTS`` (custom_multimodal_builder.py:19-498):
same constructor argument (cfg), same ``forward(x, y, return_embed, return_spatial_attn,
return_temporal_attn)`` contract, same ``state_dict`` names / shapes / order (524 entries), same
parameter initialisation, registered under the same registry name.  torch.nn modules are used ONLY as
parameter containers (names, shapes, default init); their forward() is never called -- all compute goes
through csts_amd.ops -> libcsts_hip.so.  Compute mode comes from ``cfg.CSTS_AMD.COMPUTE``:
"fp32" (exact-fp32 MFMA, parity mode) or "bf16" (bf16 MFMA, fp32 residual stream / statistics).
"""
from __future__ import annotations

import logging
import math
import weakref
from functools import partial
from typing import Dict, List, Optional

import os
import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_

from . import ops
from . import lib as L
from .registry import MODEL_REGISTRY


_SIDE_STREAMS = weakref.WeakKeyDictionary()     # model -> HIP stream of its audio trunk


def round_width(width, multiplier, min_width=1, divisor=1):
    """Channel / head rounding rule (slowfast/models/utils.py:8-21)."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


_log = logging.getLogger("csts_amd")


def resolve_compute(cfg) -> str:
    """CSTS_AMD.COMPUTE ("fp32" | "bf16" | "fp16" | "auto") together with the reference's TRAIN.MIXED_PRECISION key
    (slowfast/config/defaults.py:79; tools/train_avgaze_net.py:70,99-109,277: torch.cuda.amp.autocast = fp16 + GradScaler).
    "fp16" IS that arithmetic: IEEE-half MFMA operands with fp32 accumulation, fp32 master weights / residual stream / LayerNorm
    statistics / softmax / losses (what autocast keeps in fp32), dynamic loss scaling with skipped steps (optim.FusedAdamW).
    "bf16" is the throughput mode the benchmark is quoted in (and what the shipped YAMLs name): bfloat16 operands, same fp32 set,
    no loss scaling (fp32's exponent range).  "auto" follows the key with the REFERENCE's arithmetic: False -> fp32, True -> fp16.
    The key is never ignored silently: the mapping is logged."""
    amd = getattr(cfg, "CSTS_AMD", None)
    compute = str(getattr(amd, "COMPUTE", "auto") if amd is not None else "auto").lower()
    mp = bool(getattr(getattr(cfg, "TRAIN", None), "MIXED_PRECISION", False))
    if compute == "auto":
        compute = "fp16" if mp else "fp32"          # the reference's own arithmetic for either value of the key
    if mp:
        if compute == "fp16":
            _log.warning("TRAIN.MIXED_PRECISION True with CSTS_AMD.COMPUTE fp16: the reference's fp16 autocast + GradScaler arithmetic "
                         "(fp32 master weights / residual stream / statistics / losses, dynamic loss scaling in the optimizer kernels)")
        elif compute == "bf16":
            _log.warning("TRAIN.MIXED_PRECISION True: the reference's fp16 autocast + GradScaler runs here as the bf16 compute mode "
                         "(fp32 master weights / residual stream / statistics / losses, no loss scaling: bf16 has fp32's exponent range); "
                         "CSTS_AMD.COMPUTE fp16 selects the reference's own fp16 arithmetic")
        else:
            _log.warning("TRAIN.MIXED_PRECISION True but CSTS_AMD.COMPUTE is %r: the explicit compute mode wins, no mixed precision", compute)
    return compute


SPLIT_STAGE3 = os.environ.get("CSTS_SPLIT_STAGE3", "0") == "1"
_HALF_STREAMS = weakref.WeakKeyDictionary()
HEAD_STREAMS = os.environ.get("CSTS_HEAD_STREAMS", "1") != "0"      # temporal fusion beside the spatial fusion on the side stream (A/B switch)


class Runtime:
    """Per-model compute configuration."""

    def __init__(self, compute: str):
        compute = str(compute).lower()
        if compute not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"CSTS_AMD.COMPUTE must be 'fp32', 'bf16' or 'fp16', got {compute!r}")
        self.name = compute
        if compute != "fp32":
            # the 16-bit type is a property of the kernel library build: bf16 -> libcsts_hip.so, fp16 -> libcsts_hip_f16.so (one
            # per process, lib.set_half raises when the other one is already loaded); the enum value L.BF16 means "16-bit"
            L.set_half(compute)
        self.compute = L.F32 if compute == "fp32" else L.BF16
        # fp16: the reference's mixed precision (torch.cuda.amp.autocast + GradScaler, tools/train_avgaze_net.py:70,99-109,277):
        # the training harness multiplies the loss by a dynamic scale and the optimizer kernels unscale / skip / update it
        self.loss_scaling = compute == "fp16"
        self.act_dt = self.compute            # dtype of GEMM inputs / q,k,v / MLP hidden
        self.stream_dt = L.F32                # residual stream, LN statistics, softmax, losses


# --------------------------------------------------------------------------- parameter containers
class _Mlp(nn.Module):
    def __init__(self, dim, hidden, dim_out):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim_out)


class _Attention(nn.Module):
    """Parameters of MultiScaleAttention (attention.py:88-116) / MultiScaleDecoderAttention (:330-361) /
    Spatial- and TemporalAttention (av_attention.py:88-117, :286-314)."""

    def __init__(self, dim, heads, kind, has_pool_q, has_pool_kv, stride_q, stride_kv):
        super().__init__()
        hd = dim // heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        if kind == "dec":
            outpad = [0 if s == 1 else s - 1 for s in stride_q]
            self.upsample_q = nn.ConvTranspose3d(hd, hd, 3, stride=tuple(stride_q), padding=1, output_padding=tuple(outpad),
                                                 groups=hd, bias=False)
            self.norm_q = nn.LayerNorm(hd)
        elif has_pool_q:
            self.pool_q = nn.Conv3d(hd, hd, 3, stride=tuple(stride_q), padding=1, groups=hd, bias=False)
            self.norm_q = nn.LayerNorm(hd)
        if has_pool_kv:
            self.pool_k = nn.Conv3d(hd, hd, 3, stride=tuple(stride_kv), padding=1, groups=hd, bias=False)
            self.norm_k = nn.LayerNorm(hd)
            self.pool_v = nn.Conv3d(hd, hd, 3, stride=tuple(stride_kv), padding=1, groups=hd, bias=False)
            self.norm_v = nn.LayerNorm(hd)


class Block(nn.Module):
    """One transformer block of the path.  kind: 'enc' (MultiScaleBlock, attention.py:165-248), 'dec'
    (MultiScaleDecoderBlock, :395-479), 'spatial' / 'temporal' (av_attention.py:373-473 / :156-250)."""

    def __init__(self, kind, dim, dim_out, heads, stride_q, stride_kv, has_pool_q, has_pool_kv, mlp_hidden, drop_path, rt):
        super().__init__()
        self.kind, self.dim, self.dim_out, self.heads = kind, dim, dim_out, heads
        self.stride_q, self.stride_kv = tuple(stride_q), tuple(stride_kv)
        self.has_pool_q, self.has_pool_kv = has_pool_q, has_pool_kv
        self.drop_prob = float(drop_path)
        self.rt = rt
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads, kind, has_pool_q, has_pool_kv, stride_q, stride_kv)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, mlp_hidden, dim_out)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)

    def _drop_scales(self, B, device, keep_masks):
        """Two independent per-sample scales (attention branch, MLP branch), drop_path of common.py:46-59.
        keep_masks: (m1, m2) 0/1 masks (tests replay the reference's draws) or ("scales", s1, s2) drawn by
        CSTS.forward for all blocks at once."""
        if self.drop_prob == 0.0 or not (self.training or keep_masks is not None):
            return None, None
        keep = 1.0 - self.drop_prob
        if keep_masks is not None and len(keep_masks) == 3:
            return keep_masks[1], keep_masks[2]
        if keep_masks is not None:
            m1, m2 = keep_masks
            m1, m2 = m1.to(device=device, dtype=torch.float32), m2.to(device=device, dtype=torch.float32)
        else:
            m1 = torch.floor(keep + torch.rand(B, dtype=torch.float32, device=device))
            m2 = torch.floor(keep + torch.rand(B, dtype=torch.float32, device=device))
        return (m1 / keep).contiguous(), (m2 / keep).contiguous()

    def forward(self, x, thw, keep_masks=None, want_attn=False, spatial_audio_attn=False, x_add=None):
        """x_add (optional): a skip tensor to be added to x first (the decoder's `feat + en_feat`,
        custom_multimodal_builder.py:467-479): folded into norm1's kernel, which then also writes the sum."""
        rt = self.rt
        a = self.attn
        B, N, Cc = x.shape
        H = self.heads
        thw = list(thw)
        # x + f(LN(x)): LN returns an alias of x whose gradient (the residual branch) is folded into its backward kernel
        xn, x = ops.layer_norm(x.contiguous(), self.norm1.weight, self.norm1.bias, 1e-6, rt.act_dt, passthrough=True,
                               addend=x_add)
        w16 = lambda lin: getattr(lin, "_w16", None)      # bf16 shadow maintained by CSTS._refresh_w16 (bf16 mode)
        w16t = lambda lin: getattr(lin, "_w16t", None)    # its [in][out] twin (training): the data gradients run as NT GEMMs
        kvc = None
        if ops.kv_compact_ok(thw, self.stride_kv, self.has_pool_kv) and not spatial_audio_attn and not want_attn:
            # K/V pools with spatial stride >= 4 read (3/s)^2 of the token rows: k|v only for those (ops.QkvCompactFn)
            qkv, kvc = ops.qkv_compact(xn, a.qkv.weight, a.qkv.bias, thw, self.stride_kv, out_dt=rt.act_dt, compute=rt.compute,
                                       w16=w16(a.qkv), w16t=w16t(a.qkv))
        else:
            qkv = ops.linear(xn, a.qkv.weight, a.qkv.bias, out_dt=rt.act_dt, compute=rt.compute, w16=w16(a.qkv), w16t=w16t(a.qkv))
        mask_mode, mT, mHW = L.MASK_NONE, 0, 0
        if self.kind == "spatial":
            mask_mode, mT, mHW = L.MASK_SPATIAL, thw[0], thw[1] * thw[2]
        akind = "dec" if self.kind == "dec" else ("enc" if self.kind == "enc" else "plain")
        meta = (B, N, Cc, H, thw, akind, self.stride_q, self.stride_kv, self.has_pool_q or self.kind == "dec",
                self.has_pool_kv, mask_mode, mT, mHW, rt.act_dt)
        wq = getattr(a, "upsample_q", None) or getattr(a, "pool_q", None)
        nq = getattr(a, "norm_q", None)
        pk, nk = getattr(a, "pool_k", None), getattr(a, "norm_k", None)
        pv, nv = getattr(a, "pool_v", None), getattr(a, "norm_v", None)
        o, lse = ops.attention_inner(
            qkv,
            wq.weight if wq is not None else None, nq.weight if nq is not None else None, nq.bias if nq is not None else None,
            pk.weight if pk is not None else None, nk.weight if nk is not None else None, nk.bias if nk is not None else None,
            pv.weight if pv is not None else None, nv.weight if nv is not None else None, nv.bias if nv is not None else None,
            meta, kvc=kvc)
        res_up = None
        if self.kind == "dec":
            q_thw = [t * s for t, s in zip(thw, self.stride_q)]
            if ops.res_up_ok(q_thw) and x.dtype == torch.float32:
                x_res, res_up = x, (list(thw), q_thw)     # the proj GEMM's epilogue up-samples the skip itself
            else:
                x_res = ops.trilinear(x, thw, self.stride_q)
        elif self.kind == "enc" and self.has_pool_q:
            q_thw = [(t - 1) // s + 1 for t, s in zip(thw, self.stride_q)]
            x_res = ops.maxpool_skip(x, thw, self.stride_q)
        else:
            q_thw = thw
            x_res = x
        Nq = o.shape[1]
        s_attn, s_mlp = self._drop_scales(B, x.device, keep_masks)
        x1 = ops.linear(o, a.proj.weight, a.proj.bias, residual=x_res, row_scale=s_attn, rows_per_scale=Nq, out_dt=L.F32,
                        compute=rt.compute, w16=w16(a.proj), w16t=w16t(a.proj), res_up=res_up)
        if self.dim != self.dim_out:        # norm2's output feeds fc1 AND the skip projection (attention.py:243-246)
            xn2, xn2b, x1 = ops.layer_norm(x1, self.norm2.weight, self.norm2.bias, 1e-6, rt.act_dt, passthrough=True, fanout=True)
            base = ops.linear(xn2b, self.proj.weight, self.proj.bias, out_dt=L.F32, compute=rt.compute, w16=w16(self.proj),
                              w16t=w16t(self.proj))
        else:
            xn2, x1 = ops.layer_norm(x1, self.norm2.weight, self.norm2.bias, 1e-6, rt.act_dt, passthrough=True)
            base = x1
        out = ops.mlp(xn2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=base,
                      row_scale=s_mlp, rows_per_scale=Nq, act_dt=rt.act_dt, out_dt=L.F32, compute=rt.compute,
                      w16_1=w16(self.mlp.fc1), w16_2=w16(self.mlp.fc2), w16t_1=w16t(self.mlp.fc1), w16t_2=w16t(self.mlp.fc2))
        extra = None
        if spatial_audio_attn:       # av_attention.py:360-370: (per-head rescaled audio->pixel map, its head mean per token)
            T, HW = thw[0], thw[1] * thw[2]
            wmap, aa = ops.audio_attn(qkv, T, HW, Cc, H)     # differentiable w.r.t. qkv (train mode back-propagates through it)
            extra = (aa.reshape(B, H, T, thw[1], thw[2]), wmap)
        elif want_attn:
            with torch.no_grad():
                extra = ops.attention_probs(qkv.detach(), B, N, Cc, H, lse, mask_mode, mT, mHW)
        return out, q_thw, extra


class _PatchEmbed(nn.Module):
    def __init__(self, dim_in, dim_out, kernel, stride, padding):
        super().__init__()
        self.proj = nn.Conv3d(dim_in, dim_out, kernel_size=tuple(kernel), stride=tuple(stride), padding=tuple(padding))
        self.kernel, self.stride, self.padding = tuple(kernel), tuple(stride), tuple(padding)


@MODEL_REGISTRY.register()
class CSTS(nn.Module):
    """Audio-visual MViT with spatial/temporal fusion and transformer decoder (gaze anticipation)."""

    def __init__(self, cfg):
        super().__init__()
        assert cfg.DATA.TRAIN_CROP_SIZE == cfg.DATA.TEST_CROP_SIZE        # custom_multimodal_builder.py:28
        self.cfg = cfg
        if cfg.MVIT.NORM != "layernorm":
            raise NotImplementedError("Only supports layernorm.")          # :63
        if cfg.MVIT.CLS_EMBED_ON or not cfg.MVIT.SEP_POS_EMBED or cfg.MVIT.PATCH_2D or cfg.MVIT.POOL_FIRST \
                or cfg.MVIT.MODE != "conv" or cfg.MVIT.DROPOUT_RATE > 0 or cfg.MVIT.NORM_STEM:
            raise NotImplementedError("csts_amd implements the configuration of the shipped CSTS YAMLs: no class "
                                      "token, separable pos-embed, 3-D patches, conv pooling, no dropout, no stem norm")
        amd = getattr(cfg, "CSTS_AMD", None)
        self.rt = Runtime(resolve_compute(cfg))
        if bool(getattr(cfg.MODEL, "ACT_CHECKPOINT", False)):
            # custom_multimodal_builder.py:154,178,214 wrap every block in fairscale's checkpoint_wrapper: a memory-for-recompute
            # trade that leaves every value unchanged.  Never ignored silently: logged, like TRAIN.MIXED_PRECISION.
            _log.warning("MODEL.ACT_CHECKPOINT True: activation checkpointing is not applied here -- the saved activations of the "
                         "largest shipped configuration (32x256^2, B=1) peak at 11.6 GiB of 288 GB, and results do not depend on it")
        self.two_streams = bool(getattr(amd, "TWO_STREAMS", True)) if amd is not None else True
        rt = self.rt
        S, T = cfg.DATA.TRAIN_CROP_SIZE, cfg.DATA.NUM_FRAMES
        self.patch_stride = list(cfg.MVIT.PATCH_STRIDE)
        embed_dim, num_heads, depth = cfg.MVIT.EMBED_DIM, cfg.MVIT.NUM_HEADS, cfg.MVIT.DEPTH
        mlp_ratio = cfg.MVIT.MLP_RATIO
        dpr_rate = cfg.MVIT.DROPPATH_RATE
        self.spatial_audio_attn = cfg.MVIT.SPATIAL_AUDIO_ATTN
        self.patch_embed = _PatchEmbed(cfg.DATA.INPUT_CHANNEL_NUM[0], embed_dim, cfg.MVIT.PATCH_KERNEL, cfg.MVIT.PATCH_STRIDE,
                                       cfg.MVIT.PATCH_PADDING)
        self.patch_embed_audio = _PatchEmbed(1, embed_dim, cfg.MVIT.PATCH_KERNEL, cfg.MVIT.PATCH_STRIDE,
                                             cfg.MVIT.PATCH_PADDING)
        self.input_dims = [T, S, S]
        assert self.input_dims[1] == self.input_dims[2]                   # :84
        self.patch_dims = [self.input_dims[i] // self.patch_stride[i] for i in range(3)]
        pd = self.patch_dims
        self.pos_embed_spatial = nn.Parameter(torch.zeros(1, pd[1] * pd[2], embed_dim))
        self.pos_embed_temporal = nn.Parameter(torch.zeros(1, pd[0], embed_dim))
        self.pos_embed_spatial_audio = nn.Parameter(torch.zeros(1, pd[1] * pd[2], embed_dim))
        self.pos_embed_temporal_audio = nn.Parameter(torch.zeros(1, pd[0], embed_dim))

        # ---- video encoder geometry (:115-180)
        dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
        for i, m in cfg.MVIT.DIM_MUL:
            dim_mul[int(i)] = m
        for i, m in cfg.MVIT.HEAD_MUL:
            head_mul[int(i)] = m
        stride_q = [[] for _ in range(depth)]
        for row in cfg.MVIT.POOL_Q_STRIDE:
            stride_q[int(row[0])] = list(row[1:])
        if cfg.MVIT.POOL_KV_STRIDE_ADAPTIVE is not None:                   # :136-142 (also mutates cfg, like the reference)
            skv = list(cfg.MVIT.POOL_KV_STRIDE_ADAPTIVE)
            cfg.MVIT.POOL_KV_STRIDE = []
            for i in range(depth):
                if len(stride_q[i]) > 0:
                    skv = [max(skv[d] // stride_q[i][d], 1) for d in range(3)]
                cfg.MVIT.POOL_KV_STRIDE.append([i] + skv)
        stride_kv = [[] for _ in range(depth)]
        for row in cfg.MVIT.POOL_KV_STRIDE:
            stride_kv[int(row[0])] = list(row[1:])
        if cfg.MVIT.POOL_KVQ_KERNEL is None or list(cfg.MVIT.POOL_KVQ_KERNEL) != [3, 3, 3]:
            raise NotImplementedError("csts_amd pools with the 3x3x3 kernels of the shipped YAMLs (POOL_KVQ_KERNEL)")
        dpr = torch.linspace(0, dpr_rate, depth).tolist()
        self.blocks = nn.ModuleList()
        heads, dim = num_heads, embed_dim
        for i in range(depth):
            heads = round_width(heads, head_mul[i])
            dim = round_width(dim, dim_mul[i], divisor=heads)
            dim_out = round_width(dim, dim_mul[i + 1], divisor=round_width(heads, head_mul[i + 1]))
            has_q = len(stride_q[i]) > 0
            has_kv = len(stride_kv[i]) > 0
            self.blocks.append(Block("enc", dim, dim_out, heads, stride_q[i] if has_q else (1, 1, 1),
                                     stride_kv[i] if has_kv else (1, 1, 1), has_q, has_kv, int(dim * mlp_ratio), dpr[i], rt))
        # ---- audio encoder (:184-216)
        a_dim, a_out, a_heads = [96, 192, 384, 768], [192, 384, 768, 768], [1, 2, 4, 8]
        a_sq = [None, (1, 2, 2), (1, 2, 2), (1, 2, 2)]
        a_skv = [(1, 8, 8), (1, 4, 4), (1, 2, 2), (1, 1, 1)]
        self.blocks_audio = nn.ModuleList([
            Block("enc", a_dim[i], a_out[i], a_heads[i], a_sq[i] or (1, 1, 1), a_skv[i], a_sq[i] is not None, True,
                  int(a_dim[i] * mlp_ratio), 0.0, rt) for i in range(4)])
        token_dim = self.blocks[-1].dim_out
        if "nce" in cfg.MODEL.LOSS_FUNC:                                   # :221-224
            self.vision_proj = nn.Linear(token_dim, 256)
            self.audio_proj = nn.Linear(token_dim, 256)
        fk = (1, pd[1] // 8, pd[2] // 8) if getattr(amd, "FUSION_KERNEL_FROM_GRID", False) else (1, 8, 8)
        self.vision_pool = nn.Conv3d(token_dim, token_dim, kernel_size=fk, stride=1)     # :227-229
        self.audio_pool = nn.Conv3d(token_dim, token_dim, kernel_size=fk, stride=1)
        self.audio_pool2 = nn.Conv3d(token_dim, token_dim, kernel_size=fk, stride=1)
        self.temporal_fusion = Block("temporal", token_dim, token_dim, heads, (1, 1, 1), (1, 1, 1), False, False,
                                     int(token_dim * mlp_ratio), 0.0, rt)
        self.spatial_fusion = Block("spatial", token_dim, token_dim, heads, (1, 1, 1), (1, 1, 1), False, False,
                                    int(token_dim * mlp_ratio), 0.0, rt)
        # ---- decoder (:271-299); MLP hidden = 4 * dim_out (attention.py:444)
        d_in, d_out, d_heads = [768, 768, 384, 192], [768, 384, 192, 96], [8, 4, 4, 2]
        d_sq = [(1, 2, 2), (1, 2, 2), (1, 2, 2), (2, 1, 1)]
        d_skv = [(1, 2, 2), (1, 4, 4), (1, 8, 8), (1, 16, 16)]
        for i in range(4):
            setattr(self, f"decode_block{i + 1}",
                    Block("dec", d_in[i], d_out[i], d_heads[i], d_sq[i], d_skv[i], True, True, int(d_out[i] * mlp_ratio), 0.0, rt))
        self.classifier = nn.Conv3d(96, 1, kernel_size=1)
        # ---- init (:304-325): trunc_normal(.02) pos-embeds and Linear weights, LN 1/0; convs keep torch default
        for p_ in (self.pos_embed_spatial, self.pos_embed_temporal, self.pos_embed_spatial_audio, self.pos_embed_temporal_audio):
            trunc_normal_(p_, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _refresh_w16(self, defer_twins=False):
        """bf16 shadows of every block Linear weight (fp32 masters stay the nn.Parameters): one multi-tensor cast per
        forward in training mode (weights change every step; also what a captured HIP graph replays), on demand in eval.
        defer_twins: the [in][out] twins (read by backward only) are NOT refreshed here; the transpose set is returned and the
        caller refreshes it where it costs nothing (forward_trunk: on the side stream, beside the forward pass)."""
        if self.rt.compute != L.BF16:
            return None
        lins = getattr(self, "_w16_lins", None)
        if lins is None:
            lins = [m for blk in self.modules() if isinstance(blk, Block)
                    for m in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2, getattr(blk, "proj", None)) if m is not None]
            self._w16_lins = lins
            # + the three (1,8,8) fusion convs (37.7 M parameters each): their skinny GEMMs are pure weight streams, and a
            # bf16 shadow halves the bytes of the forward and of the data gradient (same values: the GEMM rounds the fp32
            # weights to bf16 while staging anyway).  No [in][out] twins for these.
            self._w16_all = lins + [m for m in (getattr(self, n, None) for n in ("vision_pool", "audio_pool", "audio_pool2"))
                                    if m is not None]
        shadowed = self._w16_all
        stale = [l for l in shadowed if getattr(l, "_w16", None) is None or l._w16.device != l.weight.device]
        for l in stale:
            l._w16 = torch.empty(l.weight.shape, dtype=L.half_dtype(), device=l.weight.device)
            l._w16_ver = -1
        # With csts_amd.optim.FusedAdamW the optimizer kernel itself rewrites the shadows (w16_external): only an
        # out-of-band weight change (load_state_dict, manual edit: bumps _version) needs a refresh here.
        if (self.training and not getattr(self, "w16_external", False)) or any(l._w16_ver != l.weight._version for l in shadowed):
            with torch.no_grad():
                torch._foreach_copy_([l._w16 for l in shadowed], [l.weight.detach() for l in shadowed])
            for l in shadowed:
                l._w16_ver = l.weight._version
        # [in][out] twins for the data-gradient GEMMs: one multi-tensor transpose per forward that records a backward graph
        # (refreshed in the same forward whose backward reads them: never stale)
        if torch.is_grad_enabled() and ops.USE_W16T:
            ts = getattr(self, "_w16t_set", None)
            if ts is None or ts.pairs[0][0].device != lins[0]._w16.device or any(l._w16 is not p_[0] for l, p_ in zip(lins, ts.pairs)):
                for l in lins:
                    l._w16t = torch.empty(l.weight.shape[1], l.weight.shape[0], dtype=L.half_dtype(), device=l.weight.device)
                ts = self._w16t_set = ops._TransposeSet([(l._w16, l._w16t) for l in lins])
            if defer_twins:
                return ts
            ts.refresh()
        return None

    def _draw_drop_paths(self, B, device):
        """All stochastic-depth scales of one forward in ONE draw (4 small kernels instead of 4 per block branch):
        row 2i / 2i+1 = attention / MLP branch of the i-th block with a non-zero rate (common.py:46-59:
        floor(keep + U[0,1)) / keep per sample).  Uses torch's Philox generator (graph-safe)."""
        named = [(f"blocks.{i}", b) for i, b in enumerate(self.blocks)] + \
                [(f"blocks_audio.{i}", b) for i, b in enumerate(self.blocks_audio)]
        named = [(n, b) for n, b in named if b.drop_prob > 0.0]
        if not named:
            return {}
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep.device != device:
            keep = torch.tensor([1.0 - b.drop_prob for _, b in named for _ in (0, 1)], dtype=torch.float32,
                                device=device).unsqueeze(1)
            self._dp_keep = keep
        scales = torch.floor(keep + torch.rand(keep.shape[0], B, dtype=torch.float32, device=device)) / keep
        return {n: ("scales", scales[2 * i], scales[2 * i + 1]) for i, (n, _) in enumerate(named)}

    def _audio_stream(self):
        # kept outside the module's attributes: a HIP stream can be neither pickled nor deep-copied
        st = _SIDE_STREAMS.get(self)
        if st is None:
            st = _SIDE_STREAMS[self] = torch.cuda.Stream()
        return st

    def head_parameters(self):
        """Parameters of everything after the encoder trunks (fusion convs and blocks, decoder, classifier, EgoNCE
        projections): the first gradient bucket of csts_amd.train.SegmentedTrainStep (complete when the decoder / fusion
        backward is, 60 % of all gradient bytes)."""
        mods = [self.vision_pool, self.audio_pool, self.audio_pool2, self.temporal_fusion, self.spatial_fusion,
                self.decode_block1, self.decode_block2, self.decode_block3, self.decode_block4, self.classifier]
        mods += [getattr(self, n) for n in ("vision_proj", "audio_proj") if hasattr(self, n)]
        return [p for m_ in mods for p in m_.parameters()]

    def early_trunk_parameters(self, k):
        """Parameters of the video trunk in front of block k (patch embedding, position embeddings, blocks 0 .. k-1): the LAST
        gradient bucket of a three-segment SegmentedTrainStep.  k = 3 (the 96- and 192-channel stages) is 1.3 M parameters
        of the trunks' 44.6 M but a third of their backward time, so the rest of the trunk gradients travel underneath it."""
        assert 0 < k < len(self.blocks)
        ps = [self.pos_embed_spatial, self.pos_embed_temporal] + list(self.patch_embed.parameters())
        for b in list(self.blocks)[:k]:
            ps += list(b.parameters())
        return ps

    @torch.jit.ignore
    def no_weight_decay(self):
        """custom_multimodal_builder.py:327-341."""
        if self.cfg.MVIT.ZERO_DECAY_POS_CLS:
            return {"pos_embed_spatial", "pos_embed_temporal", "pos_embed_class"}
        return {}

    # ------------------------------------------------------------------ forward (custom_multimodal_builder.py:343-498)
    def forward(self, x, y, return_embed=False, return_spatial_attn=False, return_temporal_attn=False, keep_masks=None,
                boundary=None):
        """boundary (optional, csts_amd.train.SegmentedTrainStep): callable applied to the list of tensors that cross from
        the encoder trunks to the fusion / decoder head [video tokens, audio tokens, the four encoder features the decoder
        re-uses]; it may return detached stand-ins, which cuts the autograd graph into a trunk part and a head part.  A
        boundary with an attribute `trunk_cut` = k > 0 and a method `inner(tensors)` is also applied to [video tokens] in front
        of video block k: a second cut, early video trunk | rest of the trunks (see early_trunk_parameters)."""
        inpt = x[0]
        if not inpt.is_cuda:
            raise L.CstsError("CSTS (csts_amd) runs on MI355X only: inputs must be GPU tensors; there is no CPU fallback")
        km = keep_masks or {}
        if self.training and keep_masks is None:
            km = self._draw_drop_paths(inpt.shape[0], inpt.device)
        twins = self._refresh_w16(defer_twins=self.two_streams and HEAD_STREAMS)
        feats, geo = self.forward_trunk(inpt, y, km, boundary, twins=twins)
        if boundary is not None:
            feats = boundary(feats)
        return self.forward_head(feats, geo, km, return_embed, return_spatial_attn, return_temporal_attn)

    def _run_halves(self, xt, thw, km, blocks, names, main):
        """blocks over the two halves of the batch, the second half on its own stream (autograd replays its backward there)."""
        B = xt.shape[0]
        h = B // 2
        st2 = _HALF_STREAMS.get(self)
        if st2 is None:
            st2 = _HALF_STREAMS[self] = torch.cuda.Stream()

        def km_half(v, lo, hi):
            if v is None:
                return None
            return tuple(e if isinstance(e, str) else e[lo:hi] for e in v)

        def chain(t, lo, hi):
            shape = list(thw)
            for blk, nm in zip(blocks, names):
                t, shape, _ = blk(t, shape, km_half(km.get(nm), lo, hi))
            return t, shape

        st2.wait_stream(main)
        xt.record_stream(st2)
        with torch.cuda.stream(st2):
            xb, _ = chain(xt[h:], h, B)
        xa, shape = chain(xt[:h], 0, h)
        main.wait_stream(st2)
        xb.record_stream(main)
        return torch.cat([xa, xb], dim=0), shape

    def forward_trunk(self, inpt, y, km, boundary=None, twins=None):
        """Patch embeddings + the video and audio encoders (custom_multimodal_builder.py:346-411).  twins: the weight-twin
        transpose set whose refresh _refresh_w16 left to this function."""
        rt = self.rt
        pe, pa = self.patch_embed, self.patch_embed_audio
        audio_embed = lambda: ops.patch_embed(y.float(), pa.proj.weight, pa.proj.bias, self.pos_embed_spatial_audio,
                                              self.pos_embed_temporal_audio, pa.kernel, pa.stride, pa.padding, rt.act_dt, rt.compute)
        early_side = self.two_streams and HEAD_STREAMS          # the audio patch embedding and the twin refresh start the side stream
        if twins is not None and not early_side:
            twins.refresh()
        yt = None if early_side else audio_embed()
        xt = ops.patch_embed(inpt.float(), pe.proj.weight, pe.proj.bias, self.pos_embed_spatial, self.pos_embed_temporal,
                             pe.kernel, pe.stride, pe.padding, rt.act_dt, rt.compute)
        T, H, W = self.patch_dims
        thw, thw_a = [T, H, W], [T, H, W]
        # encoder features the decoder re-uses (:384,389,396,403): each goes through ops.tap, so that its two gradients
        # (next block + decoder skip) meet in one kernel that also leaves the bf16 copy the next GEMMs read
        xt, keep = ops.tap(xt, rt.compute)
        inter = [keep]
        inter_thw = [list(thw)]

        def run(t, shape, blocks, names):
            for blk, nm in zip(blocks, names):
                t, shape, _ = blk(t, shape, km.get(nm))
            return t, shape

        vb, ab = self.blocks, self.blocks_audio
        vn = [f"blocks.{i}" for i in range(len(vb))]
        an = [f"blocks_audio.{i}" for i in range(len(ab))]
        # The reference interleaves video / audio stages (:387-411) but the two trunks are data-independent until the
        # fusion: the audio trunk runs on its own HIP stream (autograd replays its backward on that stream too), so
        # its latency-bound kernels fill CUs the video trunk leaves idle.
        main = torch.cuda.current_stream()
        side = self._audio_stream() if self.two_streams else None
        if side is not None:
            side.wait_stream(main)
            if yt is not None:
                yt.record_stream(side)  # allocated on `main`, read on `side`: keep the allocator from recycling it early
            with torch.cuda.stream(side):
                if yt is None:
                    yt = audio_embed()
                yt, thw_a = run(yt, thw_a, ab, an)
                if twins is not None and early_side:
                    twins.refresh()     # read by backward only: after the audio trunk, beside the rest of the video trunk
        cut_at = int(getattr(boundary, "trunk_cut", 0) or 0)
        B = xt.shape[0]
        split = SPLIT_STAGE3 and side is not None and B >= 2 and B % 2 == 0 and len(vb) == 16
        i = 0
        while i < len(vb):
            blk, nm = vb[i], vn[i]
            if split and i == 4:
                # EXPERIMENT (CSTS_SPLIT_STAGE3=1): blocks 4..13 (384 channels, 2048 tokens per clip: every kernel one latency-bound
                # round) as two half-batch chains on two streams
                xt, thw = self._run_halves(xt, thw, km, vb[4:14], vn[4:14], main)
                i = 14
                xt, keep = ops.tap(xt, rt.compute)
                inter.append(keep)
                inter_thw.append(list(thw))
                continue
            if cut_at and i == cut_at:
                xt = boundary.inner([xt])[0]
            xt, thw, _ = blk(xt, thw, km.get(nm))
            if i in (0, 2, 13):                  # the encoder features the decoder re-uses (:389,396,403)
                xt, keep = ops.tap(xt, rt.compute)
                inter.append(keep)
                inter_thw.append(list(thw))
            i += 1
        if side is not None:
            main.wait_stream(side)
            yt.record_stream(main)
        else:
            yt, thw_a = run(yt, thw_a, ab, an)
        return [xt, yt] + inter, (list(thw), list(thw_a), inter_thw)

    def forward_head(self, feats, geo, km, return_embed=False, return_spatial_attn=False, return_temporal_attn=False):
        """Spatial / temporal fusion, re-weighting, decoder, classifier, embeddings (custom_multimodal_builder.py:413-498)."""
        rt = self.rt
        xt, yt = feats[0], feats[1]
        thw, thw_a, inter_thw = geo
        inter = list(zip(feats[2:], inter_thw))

        # ---- spatial fusion (:415-432)
        B, Nv, Cc = xt.shape
        Tn, HW = thw[0], thw[1] * thw[2]
        HWa = thw_a[1] * thw_a[2]
        c16 = lambda m: getattr(m, "_w16", None) if rt.compute == L.BF16 else None     # bf16 shadow (CSTS._refresh_w16)
        def temporal_branch(x_t):                               # :435-451
            x_tmp = ops.fusion_conv(x_t, self.vision_pool.weight, self.vision_pool.bias, Tn, HW, rt.act_dt, rt.compute,
                                    w16=c16(self.vision_pool))
            y_tmp = ops.fusion_conv(yt, self.audio_pool2.weight, self.audio_pool2.bias, thw_a[0], HWa, rt.act_dt, rt.compute,
                                    w16=c16(self.audio_pool2))
            return self.temporal_fusion(ops.cat_tokens(x_tmp, y_tmp), (2, 2, 2), want_attn=return_temporal_attn)

        # The temporal fusion (16 tokens per clip: every kernel of it is one latency-bound round trip) does not depend on the spatial
        # one unless SPATIAL_AUDIO_ATTN re-weights its input (:438-440): it runs on the side stream beside the spatial fusion, and
        # autograd replays its backward there too.
        main = torch.cuda.current_stream()
        side = self._audio_stream() if (self.two_streams and HEAD_STREAMS and not self.spatial_audio_attn) else None
        if side is not None:
            side.wait_stream(main)
            xt.record_stream(side)
            yt.record_stream(side)
            with torch.cuda.stream(side):
                av_t, _, t_extra = temporal_branch(xt)
        y_sp = ops.fusion_conv(yt, self.audio_pool.weight, self.audio_pool.bias, thw_a[0], HWa, rt.act_dt, rt.compute,
                               w16=c16(self.audio_pool))
        av_sp = ops.cat_tokens(xt, y_sp)
        av_sp, _, sp_extra = self.spatial_fusion(av_sp, thw, want_attn=return_spatial_attn,
                                                 spatial_audio_attn=self.spatial_audio_attn)
        x_spatial, _ = ops.split_tokens(av_sp, Nv)        # (y_spatial, the audio half, is unused in the reference too: :432)
        # ---- temporal fusion (:435-451)
        if side is not None:
            main.wait_stream(side)
            av_t.record_stream(main)
            if torch.is_tensor(t_extra):
                t_extra.record_stream(main)
        else:
            x_t = xt
            if self.spatial_audio_attn:                         # :438-440
                x_t = ops.row_weight(xt, sp_extra[1])
            av_t, _, t_extra = temporal_branch(x_t)
        # ---- re-weight (:454-461)
        x_w, y_w = ops.split_tokens(av_t, Tn)
        x_rw = ops.reweight(x_spatial, x_w, Tn, HW)
        y_rw = ops.reweight(yt, y_w, thw_a[0], HWa)
        # ---- embeddings (:493-497) early, on the side stream beside the decoder (their EgoNCE backward runs there too)
        emb = None
        if return_embed:
            embed = lambda: (ops.linear(ops.token_mean(x_rw), self.vision_proj.weight, self.vision_proj.bias, out_dt=L.F32, compute=L.F32),
                             ops.linear(ops.token_mean(y_rw), self.audio_proj.weight, self.audio_proj.bias, out_dt=L.F32, compute=L.F32))
            eside = self._audio_stream() if (self.two_streams and HEAD_STREAMS) else None
            if eside is not None:
                eside.wait_stream(main)
                x_rw.record_stream(eside)
                y_rw.record_stream(eside)
                with torch.cuda.stream(eside):
                    emb = embed()
            else:
                emb = embed()
        # ---- decoder (:466-475)
        feat, dthw = x_rw, list(thw)
        skip = None                              # feat = feat + inter[...] (:469-475) happens inside the next block's norm1
        for i in range(4):
            blk = getattr(self, f"decode_block{i + 1}")
            feat, dthw, _ = blk(feat, dthw, km.get(f"decode_block{i + 1}"), x_add=skip)
            skip = inter[-1 - i][0] if i < 3 else None
        # ---- head (:476-481)
        en, en_thw = inter[0]
        logits = ops.classifier_head(feat, en, self.classifier.weight, self.classifier.bias, en_thw, rt.compute)

        if not return_embed and not return_spatial_attn and not return_temporal_attn:
            return logits
        if not return_embed:
            out = [logits]
            if return_spatial_attn:     # with SPATIAL_AUDIO_ATTN the block yields the audio->pixel map instead (the reference
                out.append(sp_extra[0] if isinstance(sp_extra, tuple) else sp_extra)   # raises NameError there, :485-491)
            if return_temporal_attn:
                out.append(t_extra)
            return out
        v_emb, a_emb = emb
        if self.two_streams and HEAD_STREAMS:
            main.wait_stream(self._audio_stream())
            v_emb.record_stream(main)
            a_emb.record_stream(main)
        return [logits, v_emb, a_emb]
