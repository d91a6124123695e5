"""Thin training-step harness: this package's counterpart of tools/train_avgaze_net.py:25-155 (reference Python
never ships to the GPU box).  It reproduces the timed iteration -- LR set, forward, frame_softmax + KLDiv +
LOSS_ALPHA * EgoNCE, backward (+ RCCL gradient all-reduce), L2 clip 1.0, AdamW -- without the reference's
per-iteration host syncs (isnan / .item() / metric gathers run only when asked).  Optimizer and LR schedule are
stock torch (a fused device optimizer is a "next" row)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import distributed as du
from . import losses
from .distributed import GradAllReduce


def get_lr_at_epoch(cfg, cur_epoch: float) -> float:
    """Cosine schedule with optional linear warm-up (slowfast/utils/lr_policy.py:9-53)."""
    s = cfg.SOLVER
    assert s.LR_POLICY == "cosine", "only the cosine policy of the CSTS YAMLs is implemented"

    def cosine(e):
        offset = s.WARMUP_EPOCHS if s.COSINE_AFTER_WARMUP else 0.0
        assert s.COSINE_END_LR < s.BASE_LR
        return s.COSINE_END_LR + (s.BASE_LR - s.COSINE_END_LR) * (math.cos(math.pi * (e - offset) / (s.MAX_EPOCH - offset)) + 1.0) * 0.5

    lr = cosine(cur_epoch)
    if cur_epoch < s.WARMUP_EPOCHS:
        lr_end = cosine(s.WARMUP_EPOCHS)
        alpha = (lr_end - s.WARMUP_START_LR) / s.WARMUP_EPOCHS
        lr = cur_epoch * alpha + s.WARMUP_START_LR
    return lr


def set_lr(optimizer, new_lr: float):
    for g in optimizer.param_groups:
        if isinstance(g["lr"], torch.Tensor):
            g["lr"].fill_(new_lr)        # device-resident lr (graph-capturable optimizer)
        else:
            g["lr"] = new_lr


def construct_optimizer(model, cfg, capturable: bool = False, device_fused: bool = True):
    """AdamW(eps 1e-8) with zero weight decay on 1-D parameters and biases
    (slowfast/models/optimizer.py:11-108 for OPTIMIZING_METHOD adamw, ZERO_WD_1D_PARAM).
    On the GPU this is csts_amd.optim.FusedAdamW: clip_grad_norm_ (train_avgaze_net.py:105-106) + AdamW + bf16 shadow
    refresh in three launches; device_fused=False gives stock torch.optim.AdamW (the numerics cross-check)."""
    assert cfg.SOLVER.OPTIMIZING_METHOD == "adamw", "the CSTS YAMLs train with adamw"
    core = model.module if isinstance(model, GradAllReduce) else model
    skip = core.no_weight_decay() if hasattr(core, "no_weight_decay") else {}
    decay, no_decay = [], []
    for name, m in core.named_modules():
        for pn, p in m.named_parameters(recurse=False):
            if not p.requires_grad:
                continue
            full = f"{name}.{pn}" if name else pn
            if full in skip or (cfg.SOLVER.ZERO_WD_1D_PARAM and (p.dim() == 1 or full.endswith(".bias"))):
                no_decay.append(p)
            else:
                decay.append(p)
    groups = [g for g in ({"params": decay, "weight_decay": cfg.SOLVER.WEIGHT_DECAY},
                          {"params": no_decay, "weight_decay": 0.0}) if g["params"]]
    fused = all(p.is_cuda for g in groups for p in g["params"])     # one multi-tensor kernel per group on the GPU
    if fused and device_fused and not cfg.SOLVER.CLIP_GRAD_VAL:
        from .optim import FusedAdamW
        shadows = {}
        if hasattr(core, "_refresh_w16"):
            core._refresh_w16()
            shadows = {id(l.weight): l._w16 for l in getattr(core, "_w16_lins", [])}
            core.w16_external = True
        return FusedAdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-8, max_grad_norm=float(cfg.SOLVER.CLIP_GRAD_L2NORM or 0.0),
                          shadows=shadows)
    kw = {}
    if fused and capturable:      # lr lives in a device tensor so that a captured step can follow the schedule
        kw = {"capturable": True}
        lr0 = torch.tensor(float(cfg.SOLVER.BASE_LR), dtype=torch.float32, device=groups[0]["params"][0].device)
        return torch.optim.AdamW(groups, lr=lr0, eps=1e-8, weight_decay=cfg.SOLVER.WEIGHT_DECAY, fused=True, **kw)
    return torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-8, weight_decay=cfg.SOLVER.WEIGHT_DECAY, fused=fused)


def compute_loss(cfg, model, video, audio, labels_hm, keep_masks=None):
    """train_avgaze_net.py:70-93."""
    if cfg.MODEL.LOSS_FUNC == "kldiv+egonce":
        logits, v_emb, a_emb = model([video], audio, return_embed=True, keep_masks=keep_masks) \
            if keep_masks is not None else model([video], audio, return_embed=True)
        if du.is_dist():
            v_emb, a_emb = du.all_gather_with_grad([v_emb, a_emb])
        preds = losses.frame_softmax(logits, temperature=2)
        kld = losses.KLDiv()(preds, labels_hm)
        nce = losses.EgoNCE()(losses.sim_matrix(v_emb, a_emb))
        return kld + cfg.MODEL.LOSS_ALPHA * nce, kld, nce, preds
    logits = model([video], audio)
    preds = losses.frame_softmax(logits, temperature=2)
    kld = losses.get_loss_func(cfg.MODEL.LOSS_FUNC)()(preds, labels_hm)
    return kld, kld, None, preds


def train_step(cfg, model, batch: Dict[str, torch.Tensor], optimizer=None, lr: Optional[float] = None, keep_masks=None):
    """One iteration: forward + loss + backward (+ gradient all-reduce) [+ clip + AdamW when an optimizer is given].
    Returns (loss, kld, nce) as device tensors (no host sync)."""
    if optimizer is not None and lr is not None:
        set_lr(optimizer, lr)
    if optimizer is not None and hasattr(optimizer, "params"):
        optimizer.zero_grad(set_to_none=True)        # cached parameter list (nn.Module.parameters() walks the tree: ~5 ms)
    else:
        for p in model.parameters():
            p.grad = None
    loss, kld, nce, _ = compute_loss(cfg, model, batch["video"], batch["audio"], batch["labels_hm"], keep_masks)
    loss.backward()
    if isinstance(model, GradAllReduce):
        model.finish()
    if optimizer is not None:
        _clip_and_step(cfg, model, optimizer)
    return loss.detach(), kld.detach(), (nce.detach() if nce is not None else None)


def _clip_and_step(cfg, model, optimizer):
    """train_avgaze_net.py:101-109 (clip, then step); FusedAdamW clips inside its own kernels."""
    from .optim import FusedAdamW
    if not isinstance(optimizer, FusedAdamW):
        if cfg.SOLVER.CLIP_GRAD_VAL:
            torch.nn.utils.clip_grad_value_(model.parameters(), cfg.SOLVER.CLIP_GRAD_VAL)
        elif cfg.SOLVER.CLIP_GRAD_L2NORM:
            torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.CLIP_GRAD_L2NORM)
    optimizer.step()


def synthetic_batch(B: int, num_frames: int, crop: int, seed: int, device, pipeline: str = "device") -> Dict[str, torch.Tensor]:
    """Synthetic clips generated ON DEVICE following the dataset contract (SURVEY.md 8(d);
    ego4d_avgaze_forecast.py:214-221,294-335): normalised uint8 video, log-power STFT windows of 24 kHz noise
    (n_fft 511, hop 120, win 240: data/preprocess.py:276-290), 19x19-Gaussian gaze heatmaps.
    pipeline="device": raw uint8 frames / waveform / gaze points go through the HIP input pipeline (csts_amd.inputs);
    pipeline="torch": the same quantities with torch ops (torch.stft), the cross-check."""
    g = torch.Generator(device=device).manual_seed(seed)
    T, S = num_frames, crop
    if pipeline == "device" and torch.device(device).type == "cuda":
        from . import inputs
        frames = torch.randint(0, 256, (B, T, S, S, 3), generator=g, device=device, dtype=torch.uint8)
        n = 24000 * 5
        wav = 0.1 * torch.randn(B, n, generator=g, device=device) + 0.05 * torch.sin(
            2 * math.pi * 440.0 * torch.arange(n, device=device) / 24000.0)[None]
        xy = torch.rand(B, T, 2, generator=g, device=device)
        labels = torch.cat([xy, torch.zeros(B, T, 1, device=device)], dim=-1)
        frames_idx = (torch.arange(T, device=device, dtype=torch.float32) + 0.5)[None].expand(B, T)
        batch = inputs.assemble_batch(frames, wav, frames_idx, float(T), labels)
        batch["labels"] = labels.double()
        return batch
    u = torch.randint(0, 256, (B, 3, T, S, S), generator=g, device=device).float()
    video = (u / 255.0 - 0.45) / 0.225
    n = 24000 * 5
    wav = 0.1 * torch.randn(B, n, generator=g, device=device) + 0.05 * torch.sin(
        2 * math.pi * 440.0 * torch.arange(n, device=device) / 24000.0)[None]
    spec = torch.stft(wav, n_fft=511, hop_length=120, win_length=240, window=torch.hann_window(240, device=device),
                      center=True, pad_mode="constant", return_complex=True)
    logp = torch.log(spec.abs() ** 2 + 1e-6)
    cols = logp.shape[-1]
    audio = torch.empty(B, 1, T, S, S, device=device)
    for t in range(T):
        c = max(128, min(cols - 129, int(round((t + 0.5) / T * cols))))
        audio[:, 0, t] = logp[:, :S, c - S // 2:c + S // 2]
    yy, xx = torch.meshgrid(torch.arange(64.0, device=device), torch.arange(64.0, device=device), indexing="ij")
    cx = (torch.rand(B, T, generator=g, device=device) * 63).round()
    cy = (torch.rand(B, T, generator=g, device=device) * 63).round()
    dx = xx[None, None] - cx[..., None, None]
    dy = yy[None, None] - cy[..., None, None]
    k = torch.exp(-(dx ** 2 + dy ** 2) / (2 * 3.2 ** 2)) * (dx.abs() <= 9) * (dy.abs() <= 9)
    hm = k / k.sum(dim=(-1, -2), keepdim=True)
    labels = torch.stack([cx / 63, cy / 63, torch.zeros_like(cx)], dim=-1).double()
    return {"video": video, "audio": audio, "labels_hm": hm, "labels": labels}


class GraphedTrainStep:
    """The whole training iteration (forward + loss + backward + clip + AdamW) captured ONCE into a HIP graph and
    replayed: ~2000 kernel launches per step become one graph launch, so the step is bounded by the kernels, not by
    Python/ctypes dispatch.  Single-process only (the RCCL bucket all-reduce path stays eager).  Inputs are copied
    into static buffers; the learning rate lives in a device tensor (set_lr) so the schedule still applies.
    Drop-path masks are drawn by torch's graph-safe Philox generator on every replay."""

    def __init__(self, cfg, model, optimizer, example_batch, warmup: int = 2, allow_collectives: bool = False):
        # RCCL collectives can be captured too (torch's ProcessGroupNCCL supports it), but that path has only been run
        # with ONE rank here (bench.py --rehearse-dist --ddp-graph): multi-GPU runs stay eager unless asked otherwise.
        assert allow_collectives or not isinstance(model, GradAllReduce), "graph capture is single-GPU; multi-GPU runs eagerly"
        if isinstance(model, GradAllReduce):
            warmup = max(warmup, 3)          # bucket order and per-bucket streams are learnt in the first iterations
        self.cfg, self.model, self.opt = cfg, model, optimizer
        self.static = {k: example_batch[k].clone() for k in ("video", "audio", "labels_hm")}
        from . import ops
        mode = ops.GROUP_WGRADS
        if mode == "capture":
            # the warm-up steps must take the same (grouped) path as the capture: its lazily built host tables and
            # pinned buffers may not be allocated while a capture is open
            ops.GROUP_WGRADS = "always"
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._step()
            torch.cuda.synchronize()
        finally:
            ops.GROUP_WGRADS = mode

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        loss, kld, nce, _ = compute_loss(self.cfg, self.model, self.static["video"], self.static["audio"],
                                         self.static["labels_hm"])
        loss.backward()
        if isinstance(self.model, GradAllReduce):
            self.model.finish()
        _clip_and_step(self.cfg, self.model, self.opt)
        return loss.detach(), kld.detach(), (nce.detach() if nce is not None else None)

    def run(self, batch=None, lr: Optional[float] = None):
        if batch is not None:
            for k in self.static:
                if batch[k] is not self.static[k]:
                    self.static[k].copy_(batch[k], non_blocking=True)
        if lr is not None:
            set_lr(self.opt, lr)
        core = self.model.module if isinstance(self.model, GradAllReduce) else self.model
        if hasattr(core, "_refresh_w16"):
            core._refresh_w16()              # only acts after an out-of-band weight change (load_state_dict)
        self.graph.replay()
        return self.out
