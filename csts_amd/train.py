"""Thin training-step harness: this package's counterpart of tools/train_avgaze_net.py:25-155 (reference Python
never ships to the GPU box).  It reproduces the timed iteration -- LR set, forward, frame_softmax + KLDiv +
LOSS_ALPHA * EgoNCE, backward (+ RCCL gradient all-reduce), L2 clip 1.0, AdamW -- without the reference's
per-iteration host syncs (isnan / .item() / metric gathers run only when asked).  The optimizer is the device-side
csts_amd.optim.FusedAdamW (clip + AdamW + bf16 shadow refresh in three launches); the LR schedule is
slowfast/utils/lr_policy.py's cosine.  Three ways to run the step: eagerly (train_step), as ONE HIP graph
(GraphedTrainStep, single GPU) or as a chain of HIP graphs with the loss and the RCCL collectives issued eagerly
between them (SegmentedTrainStep, the data-parallel step)."""
from __future__ import annotations

import contextlib
import math
import os
import time
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
    # Group membership AND order follow the reference exactly (optimizer.py:40-52), quirk included: it tests the MODULE
    # name -- `name in skip`, `name.endswith(".bias")` -- so the root-level pos_embed_* parameters (module name "") are never
    # in `skip` and keep their weight decay even with MVIT.ZERO_DECAY_POS_CLS, and only the 1-D test ever fires for biases.
    # The "optimizer_state" indices of a reference checkpoint are positions in these lists (checkpoint.py:131).
    for name, m in core.named_modules():
        for p in m.parameters(recurse=False):
            if not p.requires_grad:
                continue
            if name in skip or (cfg.SOLVER.ZERO_WD_1D_PARAM and (p.dim() == 1 or name.endswith(".bias"))):
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
            shadows = {id(l.weight): l._w16 for l in (getattr(core, "_w16_all", None) or getattr(core, "_w16_lins", []))}
            core.w16_external = True
        return FusedAdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-8, max_grad_norm=float(cfg.SOLVER.CLIP_GRAD_L2NORM or 0.0),
                          shadows=shadows, loss_scaling=bool(getattr(getattr(core, "rt", None), "loss_scaling", False)))
    if bool(getattr(getattr(core, "rt", None), "loss_scaling", False)):
        # fp16 compute mode: the GradScaler semantics (unscale, non-finite check, skipped step, back-off / growth) live inside
        # FusedAdamW's kernels.  A stock torch optimizer would train on unscaled fp16 gradients that underflow silently and the
        # checkpoint would carry no scaler_state: refuse instead.
        raise RuntimeError("CSTS_AMD.COMPUTE fp16 (TRAIN.MIXED_PRECISION) needs the device-fused optimizer: it carries the dynamic loss "
                           "scaling.  Unset SOLVER.CLIP_GRAD_VAL (use CLIP_GRAD_L2NORM), keep the model on the GPU and "
                           "device_fused=True, or train in CSTS_AMD.COMPUTE bf16 / fp32.")
    kw = {}
    if fused and capturable:      # lr lives in a device tensor so that a captured step can follow the schedule
        kw = {"capturable": True}
        lr0 = torch.tensor(float(cfg.SOLVER.BASE_LR), dtype=torch.float32, device=groups[0]["params"][0].device)
        return torch.optim.AdamW(groups, lr=lr0, eps=1e-8, weight_decay=cfg.SOLVER.WEIGHT_DECAY, fused=True, **kw)
    return torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-8, weight_decay=cfg.SOLVER.WEIGHT_DECAY, fused=fused)


class LossScaler:
    """The reference's `scaler` object (torch.cuda.amp.GradScaler, tools/train_avgaze_net.py:277) for the fp16 compute mode: the
    state lives on the device inside FusedAdamW (its kernels unscale, skip and update); this view gives the harness scale() and
    the checkpoint code state_dict() / load_state_dict() ("scaler_state", slowfast/utils/checkpoint.py:133-134,351-352)."""

    def __init__(self, optimizer):
        self.opt = optimizer

    def scale(self, loss):
        return loss * self.opt.loss_scale

    def get_scale(self) -> float:
        return float(self.opt.loss_scale)

    def state_dict(self):
        return self.opt.scaler_state_dict()

    def load_state_dict(self, sd):
        self.opt.load_scaler_state_dict(sd)


def scaler_of(optimizer):
    """LossScaler when the optimizer carries dynamic loss scaling (CSTS_AMD.COMPUTE fp16), else None."""
    return LossScaler(optimizer) if getattr(optimizer, "loss_scale", None) is not None else None


def _dcopy(dst, src):
    """dst <- src, both contiguous and of one dtype: one library launch on the GPU (csts_axpby), torch elsewhere"""
    if dst.is_cuda and dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.dtype in (torch.float32, torch.bfloat16, torch.float16):
        from . import ops
        ops.cast_into(dst, src)
    else:
        dst.copy_(src)


def backward_scaled(loss, optimizer):
    """scaler.scale(loss).backward() (train_avgaze_net.py:99) -- plain loss.backward() without loss scaling."""
    sc = getattr(optimizer, "loss_scale", None) if optimizer is not None else None
    (loss if sc is None else loss * sc).backward()


def _factored_on(cfg, core) -> bool:
    """CSTS_AMD.FACTORED_ADAMW (default on) in the 16-bit compute modes: there the update forms dY^T A on the matrix cores
    (profiles/r4_factored_adamw_ab.txt: -0.3 ms per step at b = 4); the fp32 mode's scalar form measured slower and stays off.
    CSTS_FACTORED_ADAMW=0 / 1 (environment) overrides both for same-box A/B runs."""
    env = os.environ.get("CSTS_FACTORED_ADAMW", "")
    if env in ("0", "1"):
        return env == "1"
    from . import lib as L
    return bool(getattr(getattr(cfg, "CSTS_AMD", None), "FACTORED_ADAMW", True)) and getattr(getattr(core, "rt", None), "compute", L.F32) != L.F32


def _factored_params(cfg, core, optimizer):
    """The fusion-conv weights whose gradient the optimizer can form on the fly from its rank-(B T') factors
    (FusedAdamW.set_factored; CSTS_AMD.FACTORED_ADAMW): never written to memory, never read back."""
    if not _factored_on(cfg, core) or not hasattr(optimizer, "set_factored"):
        return []
    ws = [m.weight for m in (getattr(core, n, None) for n in ("vision_pool", "audio_pool", "audio_pool2")) if m is not None]
    return [w for w in ws if w.requires_grad and w.grad is None and w.is_cuda]


def backward_with_factors(cfg, model, loss, optimizer):
    """scaler.scale(loss).backward() with the fusion-conv weight gradients left as factors for the optimizer (see
    _factored_params): FusionConvFn.backward hands (dY, A) to the sink instead of running its TN GEMM."""
    from . import ops
    core = model.module if isinstance(model, GradAllReduce) else model
    fac = [] if isinstance(model, GradAllReduce) else _factored_params(cfg, core, optimizer)
    if not fac:
        if hasattr(optimizer, "set_factored"):
            optimizer.set_factored(None)
        backward_scaled(loss, optimizer)
        return
    # "accept": FusionConvFn.backward asks BEFORE it drops dW -- a geometry the factored update does not take (more than
    # FusedAdamW.FACTORED_MAX_T token rows: large per-GPU batches) runs the ordinary TN GEMM and lands in p.grad as usual
    sink = {"params": {p.data_ptr() for p in fac}, "items": [], "accept": lambda W, bt: optimizer.factored_ok(W, bt)}
    ops.set_factor_sink(sink)
    try:
        backward_scaled(loss, optimizer)
    finally:
        ops.set_factor_sink(None)
    items = []
    for W, dy, A, _ in sink["items"]:
        bt = dy.numel() // W.shape[0]
        items.append((W, dy.reshape(bt, W.shape[0]), A.reshape(bt, -1)))
    optimizer.set_factored(items)


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
    if optimizer is not None and getattr(optimizer, "_ext_grads", None) is not None:
        # 16-bit bucket views left behind by a SegmentedTrainStep on this optimizer: this step's gradients are p.grad
        optimizer.set_external_grads(None)
    if optimizer is not None and hasattr(optimizer, "params"):
        optimizer.zero_grad(set_to_none=True)        # cached parameter list (nn.Module.parameters() walks the tree: ~5 ms)
    else:
        for p in model.parameters():
            p.grad = None
    loss, kld, nce, _ = compute_loss(cfg, model, batch["video"], batch["audio"], batch["labels_hm"], keep_masks)
    if optimizer is not None:
        backward_with_factors(cfg, model, loss, optimizer)
    else:
        backward_scaled(loss, optimizer)
    if isinstance(model, GradAllReduce):
        model.finish()
    if optimizer is not None:
        _clip_and_step(cfg, model, optimizer)
        if hasattr(optimizer, "set_factored") and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            # eager: the factors belong to THIS backward pass; a step() without a fresh backward must not reuse them.  (Under
            # capture they are the graph's static tensors and stay installed for its replays.)
            optimizer.set_factored(None)
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
        if S != 256:      # the 224^2 extension: S frequency bins x S columns around each frame, (S/4)^2 heat maps
            o = (256 - S) // 2
            batch["audio"] = batch["audio"][:, :, :, :S, o:o + S].contiguous()
            batch["labels_hm"] = inputs.gaze_heatmaps(labels, H=S // 4, W=S // 4)
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
    G = S // 4
    yy, xx = torch.meshgrid(torch.arange(float(G), device=device), torch.arange(float(G), device=device), indexing="ij")
    cx = (torch.rand(B, T, generator=g, device=device) * (G - 1)).round()
    cy = (torch.rand(B, T, generator=g, device=device) * (G - 1)).round()
    dx = xx[None, None] - cx[..., None, None]
    dy = yy[None, None] - cy[..., None, None]
    k = torch.exp(-(dx ** 2 + dy ** 2) / (2 * 3.2 ** 2)) * (dx.abs() <= 9) * (dy.abs() <= 9)
    hm = k / k.sum(dim=(-1, -2), keepdim=True)
    labels = torch.stack([cx / (G - 1), cy / (G - 1), torch.zeros_like(cx)], dim=-1).double()
    return {"video": video, "audio": audio, "labels_hm": hm, "labels": labels}


class _TrainStateSnapshot:
    """Parameters, optimizer moments / step count and the device RNG state, taken before the warm-up iterations of a graph
    capture and put back after it: warm-up runs REAL optimizer steps (lazy tables, autotuned plans and the allocator have to
    see the real thing), and a fresh or resumed run must not start two AdamW steps down the road with the schedule's
    learning rate ignored (the capture itself executes nothing)."""

    def __init__(self, model, optimizer):
        self.params = [p for p in model.parameters()]
        self.values = [p.detach().clone() for p in self.params]
        self.opt = optimizer
        if hasattr(optimizer, "exp_avg"):                       # csts_amd.optim.FusedAdamW
            self.opt_state = (optimizer.exp_avg.clone(), optimizer.exp_avg_sq.clone(), optimizer.state_t.clone(), optimizer._lr.clone())
            self.scaler = optimizer.scaler_t.clone() if getattr(optimizer, "scaler_t", None) is not None else None
        else:
            import copy
            self.opt_state = copy.deepcopy(optimizer.state_dict())
        self.rng = torch.cuda.get_rng_state() if torch.cuda.is_available() else None

    def restore(self):
        with torch.no_grad():
            torch._foreach_copy_(self.params, self.values)      # bumps the version counters: bf16 shadows refresh on the next run
        if hasattr(self.opt, "exp_avg"):
            a, b, c, d = self.opt_state
            self.opt.exp_avg.copy_(a); self.opt.exp_avg_sq.copy_(b); self.opt.state_t.copy_(c); self.opt._lr.copy_(d)
            if self.scaler is not None:
                self.opt.scaler_t.copy_(self.scaler)
        else:
            self.opt.load_state_dict(self.opt_state)
        if self.rng is not None:
            torch.cuda.set_rng_state(self.rng)
        self.values = self.opt_state = None


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
        if getattr(optimizer, "_ext_grads", None) is not None:
            optimizer.set_external_grads(None)       # bucket views of an earlier SegmentedTrainStep on this optimizer
        from . import ops
        mode = ops.GROUP_WGRADS
        if mode == "capture":
            # the warm-up steps must take the same (grouped) path as the capture: its lazily built host tables and
            # pinned buffers may not be allocated while a capture is open
            ops.GROUP_WGRADS = "always"
        snap = _TrainStateSnapshot(model, optimizer)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ops.refill_capture_pools()
            if hasattr(optimizer, "refill_capture_pool"):
                optimizer.refill_capture_pool()
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: a process group's watchdog thread may poll events while this capture is open (see SegmentedTrainStep)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = self._step()
            torch.cuda.synchronize()
        finally:
            ops.GROUP_WGRADS = mode
            snap.restore()

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        loss, kld, nce, _ = compute_loss(self.cfg, self.model, self.static["video"], self.static["audio"],
                                         self.static["labels_hm"])
        backward_with_factors(self.cfg, self.model, loss, self.opt)
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



class SegmentedTrainStep:
    """The training iteration as a CHAIN of HIP graphs -- forward | backward of the fusion/decoder head | backward of the
    encoder trunks (in one or two parts) | clip + AdamW -- with the loss and every RCCL collective issued EAGERLY between them:

        G_fwd -> [EgoNCE all-gather, losses, their backward: ~20 small launches] -> G_bwd_head -> all-reduce(head grads, async)
              -> G_bwd_trunk -> all-reduce(trunk grads, async)
              [-> G_bwd_trunk_early -> all-reduce(early video trunk grads)]  -> wait -> G_opt

    trunk_cut = k > 0 (CSTS_AMD.TRUNK_CUT, default 3) cuts the trunks once more in front of video block k: the last bucket,
    the only one whose all-reduce nothing hides, is then the 1.3 M parameters of the 96- / 192-channel stages instead of all
    44.6 M of the trunks, and the other 43 M travel underneath those stages' backward (a third of the trunk backward).

    This is the data-parallel step: the compute runs from graphs exactly as on one GPU (grouped weight gradients, deferred
    reductions, two-stream trunks included), no collective is ever captured, and the head bucket -- the three 37.7 M-parameter
    fusion convs, 60 % of all gradient bytes -- travels over xGMI underneath the whole trunk backward.  Gradients are
    averaged in two flat fp32 buffers (filled by one multi-tensor copy at the end of each backward graph) and the optimizer
    reads them through p.grad views.  With one process it degenerates to the same chain without collectives (used by
    bench.py to time forward / backward / optimizer separately)."""

    def __init__(self, cfg, model, optimizer, example_batch, warmup: int = 2, use_graphs: bool = True, loss_fn=None,
                 trunk_cut: Optional[int] = None):
        """use_graphs=False runs the same chain eagerly (any device: the bucket / collective logic is then testable with
        gloo on the CPU); loss_fn(outs, static) -> (loss, kld, nce) replaces the built-in HIP losses (tests)."""
        from . import ops
        self.cfg, self.model, self.opt = cfg, model, optimizer
        self.core = model.module if isinstance(model, GradAllReduce) else model
        self.dist = du.is_dist()
        self.use_graphs, self.loss_fn = use_graphs, loss_fn
        if isinstance(model, GradAllReduce):
            model.hooks_enabled = False               # buckets are driven from here, not from autograd hooks ...
            if ops.GROUP_WGRADS == "never":           # ... so nothing reads a gradient in the middle of backward any more:
                ops.GROUP_WGRADS = "capture"          # the grouped end-of-backward weight gradients are back
        self.static = {k: example_batch[k].clone() for k in ("video", "audio", "labels_hm")}
        core = self.core
        self.head_params = [p for p in core.head_parameters() if p.requires_grad]
        # Factor exchange for the three (1,8,8) fusion convs (ops.set_factor_sink): their weights go to the END of the head bucket,
        # outside the all-reduced prefix; the ranks all-gather dY / A (3.2 MB per conv) and form the averaged dW themselves.
        amd_ = getattr(cfg, "CSTS_AMD", None)
        self.factor_params = []
        opt_factored = hasattr(optimizer, "set_factored") and _factored_on(cfg, core)
        if use_graphs and ((self.dist and bool(getattr(amd_, "FUSION_GRAD_FACTORS", True))) or (not self.dist and opt_factored)):
            self.factor_params = [m.weight for m in (getattr(core, n, None) for n in ("vision_pool", "audio_pool", "audio_pool2"))
                                  if m is not None and m.weight.requires_grad]
        self.opt_factored = opt_factored
        fid = {id(p) for p in self.factor_params}
        self.head_params = [p for p in self.head_params if id(p) not in fid] + self.factor_params
        hid = {id(p) for p in self.head_params}
        if trunk_cut is None:
            trunk_cut = int(getattr(getattr(cfg, "CSTS_AMD", None), "TRUNK_CUT", 0) or 0)
        if not hasattr(core, "early_trunk_parameters"):
            trunk_cut = 0
        self.trunk_cut = trunk_cut
        self.early_params = [p for p in core.early_trunk_parameters(trunk_cut) if p.requires_grad] if trunk_cut else []
        hid |= {id(p) for p in self.early_params}
        self.trunk_params = [p for p in core.parameters() if p.requires_grad and id(p) not in hid]
        # gradient buckets in the order their backward segments finish
        self.buckets = [self.head_params, self.trunk_params] + ([self.early_params] if trunk_cut else [])
        self._avg = None
        # CSTS_AMD.GRAD_BUCKET_DTYPE: "fp32" (default) or "bf16" / "fp16" / "half" = the 16-bit type of the kernel library: the
        # buckets travel over xGMI in 16 bits (376 MB instead of 753 MB per step) and the optimizer kernels read them as they
        # arrive, accumulating in fp32 (csts_opt_args.grad_dt).  Gradients are still PRODUCED in fp32 (weight gradients
        # accumulate over up to 262 k tokens); one cast per bucket sits at the end of its backward graph.
        self.bucket16 = str(getattr(getattr(cfg, "CSTS_AMD", None), "GRAD_BUCKET_DTYPE", "fp32")).lower() in ("bf16", "fp16", "half")
        if self.dist:
            import torch.distributed as dist
            self._avg = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else None
            self.flat = [self._flat_for(ps) for ps in self.buckets]
            if self.bucket16:
                from . import lib as L
                self.flat16 = []
                for (f32, _), ps in zip(self.flat, self.buckets):
                    f16 = torch.zeros(f32.numel(), dtype=L.half_dtype(), device=f32.device)
                    o, views = 0, []
                    for p in ps:
                        views.append(f16[o:o + p.numel()].view_as(p))
                        o += (p.numel() + 3) // 4 * 4
                    self.flat16.append((f16, views))
                if hasattr(optimizer, "set_external_grads"):     # FusedAdamW reads the 16-bit buckets directly (fp32 accumulation)
                    optimizer.set_external_grads({id(p): v for (_, views), ps in zip(self.flat16, self.buckets)
                                                  for p, v in zip(ps, views)}, L.BF16)
        self.events = None
        self.graphs = {}
        if not use_graphs:
            return
        mode = ops.GROUP_WGRADS
        ops.GROUP_WGRADS = "always" if mode != "never" else mode     # warm-up takes the same (grouped) path as the capture
        snap = _TrainStateSnapshot(model, optimizer)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self._chain(capture=False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if self.dist:
                time.sleep(0.5)          # the watchdog drops the (finished) warm-up collectives from its list
            ops.refill_capture_pools()
            if hasattr(optimizer, "refill_capture_pool"):
                optimizer.refill_capture_pool()
            self._chain(capture=True)
            torch.cuda.synchronize()
        finally:
            ops.GROUP_WGRADS = mode
            snap.restore()

    # ------------------------------------------------------------------ pieces
    def _flat_for(self, params):
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        return flat, [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, params)]

    def _segment(self, name, fn, capture):
        """Run fn() eagerly, or capture it into graph `name` (all graphs share one memory pool: they always replay in the
        same order, so a block freed by one may be re-used by the next)."""
        if not capture:
            return fn()
        g = torch.cuda.CUDAGraph()
        pool = self.graphs["fwd"].pool() if "fwd" in self.graphs else None
        # thread_local: RCCL's watchdog thread polls the events of the eager collectives issued between the captures
        # (hipEventQuery); under the default "global" mode that call fails while ANY stream is capturing and takes the
        # process down ("operation not permitted when stream is capturing")
        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
            out = fn()
        self.graphs[name] = g
        return out

    def _all_reduce(self, k):
        import torch.distributed as dist
        flat = (self.flat16 if self.bucket16 else self.flat)[k][0]
        if k == 0 and self.factor_params:          # the fusion-conv weights sit behind the all-reduced prefix of the head bucket
            n_tail = sum((p.numel() + 3) // 4 * 4 for p in self.factor_params)
            flat = flat[:flat.numel() - n_tail]
        if self._avg is not None:
            return dist.all_reduce(flat, op=self._avg, async_op=True)
        flat /= dist.get_world_size()
        return dist.all_reduce(flat, async_op=True)

    def _loss_eager(self):
        """EgoNCE all-gather + frame_softmax + KLDiv + LOSS_ALPHA * EgoNCE on the forward graph's static outputs, and the
        backward of just that part into the static gradient buffers the head graph starts from."""
        cfg = self.cfg
        leaves = [t.detach().requires_grad_(True) for t in self.outs]
        if self.loss_fn is not None:
            loss, kld, nce = self.loss_fn(leaves, self.static)
        elif cfg.MODEL.LOSS_FUNC == "kldiv+egonce":
            logits, v_emb, a_emb = leaves
            if du.is_dist():
                v_emb, a_emb = du.all_gather_with_grad([v_emb, a_emb])
            preds = losses.frame_softmax(logits, temperature=2)
            kld = losses.KLDiv()(preds, self.static["labels_hm"])
            nce = losses.EgoNCE()(losses.sim_matrix(v_emb, a_emb))
            loss = kld + cfg.MODEL.LOSS_ALPHA * nce
        else:
            preds = losses.frame_softmax(leaves[0], temperature=2)
            kld = losses.get_loss_func(cfg.MODEL.LOSS_FUNC)()(preds, self.static["labels_hm"])
            loss, nce = kld, None
        backward_scaled(loss, self.opt)
        for buf, leaf in zip(self.douts, leaves):
            buf.copy_(leaf.grad)
        return loss.detach(), kld.detach(), (nce.detach() if nce is not None else None)

    # ------------------------------------------------------------------ the loss section as a graph (round 5)
    def _loss_graphable(self):
        return (self.use_graphs and self.loss_fn is None and self.cfg.MODEL.LOSS_FUNC == "kldiv+egonce" and len(self.outs) == 3
                and os.environ.get("CSTS_LOSS_GRAPH", "1") != "0")

    def _loss_gather(self, coll):
        """The EgoNCE embeddings of all ranks into the static buffers the loss graph reads (train_avgaze_net.py:82-83); without
        collectives (one process, or the capture pass) the local rows stand in for every rank's."""
        import torch.distributed as dist
        _, v_emb, a_emb = self.outs
        if coll and dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self._gv, v_emb.contiguous())
            dist.all_gather_into_tensor(self._ga, a_emb.contiguous())
        elif coll:                                   # gloo (tests): the list form over views
            world = dist.get_world_size()
            dist.all_gather(list(self._gv.chunk(world)), v_emb.contiguous())
            dist.all_gather(list(self._ga.chunk(world)), a_emb.contiguous())
        else:
            B = v_emb.shape[0]
            for r in range(self._gv.shape[0] // B):     # (library copies: a torch copy_ between two device buffers costs the host ~100 us here)
                _dcopy(self._gv[r * B:(r + 1) * B], v_emb)
                _dcopy(self._ga[r * B:(r + 1) * B], a_emb)

    def _loss_scatter(self, coll):
        """distributed._AllGatherWithGrad.backward on the loss graph's static gradients: sum over ranks, this rank's rows."""
        import torch.distributed as dist
        if coll:
            dist.all_reduce(self._dgv)
            dist.all_reduce(self._dga)
        B = self.douts[1].shape[0]
        r = dist.get_rank() if self.dist else 0
        _dcopy(self.douts[1], self._dgv[r * B:(r + 1) * B])
        _dcopy(self.douts[2], self._dga[r * B:(r + 1) * B])

    def _loss_body(self):
        """frame_softmax + KLDiv + LOSS_ALPHA * EgoNCE on static buffers and the backward of just that, into static gradient buffers: the
        arithmetic of _loss_eager (same ops, same order) with the all-gather and its backward all-reduce taken OUT -- they are issued
        eagerly around this graph.  ~20 small launches: issued eagerly they left ~0.5 ms of host-bound gaps per step on the device
        (tools/gap_report.py on bench.py --rehearse-dist, round 5)."""
        cfg = self.cfg
        logits = self.outs[0].detach().requires_grad_(True)
        gv, ga = self._gv.detach().requires_grad_(True), self._ga.detach().requires_grad_(True)
        preds = losses.frame_softmax(logits, temperature=2)
        kld = losses.KLDiv()(preds, self.static["labels_hm"])
        nce = losses.EgoNCE()(losses.sim_matrix(gv, ga))
        loss = kld + cfg.MODEL.LOSS_ALPHA * nce
        backward_scaled(loss, self.opt)
        self.douts[0].copy_(logits.grad)
        self._dgv.copy_(gv.grad)
        self._dga.copy_(ga.grad)
        return loss.detach(), kld.detach(), nce.detach()

    def _loss_graphed(self, capture, coll):
        import torch.distributed as dist
        if capture or not hasattr(self, "_gv"):
            world = dist.get_world_size() if self.dist else 1
            _, v_emb, a_emb = self.outs
            self._gv = torch.zeros(world * v_emb.shape[0], *v_emb.shape[1:], dtype=v_emb.dtype, device=v_emb.device)
            self._ga = torch.zeros(world * a_emb.shape[0], *a_emb.shape[1:], dtype=a_emb.dtype, device=a_emb.device)
            self._dgv, self._dga = torch.zeros_like(self._gv), torch.zeros_like(self._ga)
        self._loss_gather(coll)
        res = self._segment("loss", self._loss_body, capture)
        self._loss_scatter(coll)
        return res

    def step_eager(self, batch=None, lr: Optional[float] = None):
        """One iteration of the chain without graphs (use_graphs=False)."""
        if batch is not None:
            for k in self.static:
                self.static[k].copy_(batch[k])
        if lr is not None:
            set_lr(self.opt, lr)
        return self._chain(capture=False)

    def _chain(self, capture):
        from . import ops
        if self.dist and self.use_graphs:
            # parameter gradients are produced straight inside the flat buckets where the producer allocates them (ops._grad_buffer)
            ops.set_grad_targets([(p, v) for k, ps in enumerate(self.buckets) for p, v in zip(ps, self.flat[k][1])])
        try:
            return self._chain_body(capture)
        finally:
            ops.set_grad_targets(None)

    def _copy_into_bucket(self, k):
        """Gradients of bucket k that were not produced in place -> their slices of the flat buffer (one multi-tensor copy)."""
        dst, src = [], []
        for p, v in zip(self.buckets[k], self.flat[k][1]):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if self.bucket16:           # the bucket leaves in 16 bits: one cast at the end of this backward segment
            from . import ops
            ops.cast_into(self.flat16[k][0], self.flat[k][0])

    def _chain_body(self, capture):
        core, cfg = self.core, self.cfg
        with_embed = cfg.MODEL.LOSS_FUNC == "kldiv+egonce"
        if hasattr(self.opt, "params"):
            self.opt.zero_grad(set_to_none=True)
        else:
            for ps in self.buckets:
                for p in ps:
                    p.grad = None
        cut, cut2 = {}, {}

        class _Cuts:
            """The autograd cuts of one forward: __call__ = trunks | head, inner = early video trunk | rest of the trunks."""
            trunk_cut = self.trunk_cut

            def __call__(self_, feats):
                cut["out"] = feats
                cut["in"] = [f.detach().requires_grad_(True) for f in feats]
                return cut["in"]

            def inner(self_, feats):
                cut2["out"] = feats
                cut2["in"] = [f.detach().requires_grad_(True) for f in feats]
                return cut2["in"]

        boundary = _Cuts()

        def fwd():
            out = core([self.static["video"]], self.static["audio"], return_embed=with_embed, boundary=boundary)
            return list(out) if with_embed else [out]

        self.outs = self._segment("fwd", fwd, capture)
        if capture or not hasattr(self, "douts"):
            self.douts = [torch.zeros_like(o) for o in self.outs]
        # The capture pass issues NO collective: the graphs are only recorded there, so what the eager parts between them
        # compute is thrown away anyway -- and RCCL's watchdog thread, which polls the events of unfinished collectives
        # (hipEventQuery), then has nothing to poll while a capture is open (two different HIP capture errors came out of
        # that thread otherwise, taking the process down).  Every rank takes the same branch.
        coll = self.dist and not capture
        if self._loss_graphable():
            res = self._loss_graphed(capture, coll)
        else:
            with (du.local_only() if (self.dist and capture) else contextlib.nullcontext()):
                res = self._loss_eager()
        if capture:
            self.result = res

        def bwd_head():
            from . import ops
            sink = {"params": {p.data_ptr() for p in self.factor_params}, "items": []} if self.factor_params else None
            ops.set_factor_sink(sink)
            try:
                torch.autograd.backward(self.outs, self.douts, inputs=self.head_params + cut["in"])
            finally:
                ops.set_factor_sink(None)
            if sink is not None:
                self._factors = sink["items"]          # static tensors of the captured graph (kept alive here)
            if self.dist:
                self._copy_into_bucket(0)

        self._segment("bwd_head", bwd_head, capture)
        works = [self._all_reduce(0)] if coll else []
        works += self._exchange_factors(coll)

        def bwd_trunk():
            # with a second cut, the tensors produced in front of it (encoder features of the early stages) wait for the
            # early segment: this backward starts from the late trunk's outputs only and ends at the cut's stand-in
            late = [(o, i.grad) for o, i in zip(cut["out"], cut["in"]) if not self._from_early(o, cut2)]
            torch.autograd.backward([o for o, _ in late], [g for _, g in late], inputs=self.trunk_params + cut2.get("in", []))
            if self.dist:
                self._copy_into_bucket(1)

        self._segment("bwd_trunk", bwd_trunk, capture)
        if coll:
            works.append(self._all_reduce(1))
        if self.trunk_cut:
            def bwd_early():
                early = [(o, i.grad) for o, i in zip(cut["out"], cut["in"]) if self._from_early(o, cut2)]
                roots = cut2["out"] + [o for o, _ in early]
                grads = [t.grad for t in cut2["in"]] + [g for _, g in early]
                torch.autograd.backward(roots, grads, inputs=self.early_params)
                if self.dist:
                    self._copy_into_bucket(2)

            self._segment("bwd_trunk_early", bwd_early, capture)
            if coll:
                works.append(self._all_reduce(2))
        if self.dist:
            for w in works:
                w.wait()
            self._raw_grads = [p.grad for ps in self.buckets for p in ps]   # graph-owned: keep them alive
            self._bucket16_back()
            for k, ps in enumerate(self.buckets):
                for p, v in zip(ps, self.flat[k][1]):
                    p.grad = v
        self._finish_factors()
        self._segment("opt", lambda: _clip_and_step(cfg, self.model, self.opt), capture)
        del cut, cut2
        return res

    # ------------------------------------------------------------------ factor exchange of the fusion-conv weight gradients
    def _exchange_factors(self, coll):
        """After the head backward: all-gather (dY / W, A) of every fusion conv over the ranks (async); without collectives (one
        process, or the capture pass) the local factors stand in.  Returns the outstanding works."""
        import torch.distributed as dist
        from . import ops
        works = []
        self._gathered = []
        if not self.factor_params:
            return works
        world = dist.get_world_size() if self.dist else 1
        bufs = getattr(self, "_factor_bufs", None)
        if bufs is None:
            bufs = self._factor_bufs = {}
        for i, (W, dy, A, compute) in enumerate(self._factors):
            BT = dy.numel() // W.shape[0]
            dy2, A2 = dy.reshape(BT, W.shape[0]), A.reshape(BT, -1)
            if i not in bufs:
                bufs[i] = (torch.empty(BT, W.shape[0], dtype=dy2.dtype, device=dy2.device),
                           torch.empty(world * BT, W.shape[0], dtype=dy2.dtype, device=dy2.device),
                           torch.empty(world * BT, A2.shape[1], dtype=A2.dtype, device=A2.device))
            dys, gdy, gA = bufs[i]
            ops.cast_into(dys, dy2.contiguous(), scale=1.0 / world)          # the mean over ranks, folded into the small factor
            if coll:
                # RCCL: the flat all-gather (rank r's rows land at [r * BT, (r + 1) * BT)); gloo (tests): the list form over views
                if dist.get_backend() == "nccl":
                    works.append(dist.all_gather_into_tensor(gdy, dys, async_op=True))
                    works.append(dist.all_gather_into_tensor(gA, A2.contiguous(), async_op=True))
                else:
                    works.append(dist.all_gather(list(gdy.chunk(world)), dys, async_op=True))
                    works.append(dist.all_gather(list(gA.chunk(world)), A2.contiguous(), async_op=True))
                self._gathered.append((W, gdy, gA, compute, world * BT))
            else:       # no collective in this pass: every rank slot holds the local factors (the capture pass's results are discarded)
                for r in range(world):
                    gdy[r * BT:(r + 1) * BT].copy_(dys)
                    gA[r * BT:(r + 1) * BT].copy_(A2)
                self._gathered.append((W, gdy, gA, compute, world * BT))
        return works

    def _finish_factors(self):
        """dW = [dY_0/W; ..]^T [A_0; ..] straight into the bucket slot the optimizer reads (fp32, or the 16-bit twin)."""
        from . import ops, lib as L
        if not self.factor_params:
            return
        if self.opt_factored and all(self.opt.factored_ok(W, rows) for W, _, _, _, rows in self._gathered):
            # few enough token rows (W * B * T' <= 64): the optimizer forms dW on the fly from the gathered factors
            self.opt.set_factored([(W, gdy, gA) for W, gdy, gA, _, _ in self._gathered])
            return
        if hasattr(self.opt, "set_factored"):
            self.opt.set_factored(None)
        tgt = self.averaged_grads() if self.dist else None
        for W, gdy, gA, compute, rows in self._gathered:
            if tgt is not None:
                out = tgt[W].view(W.shape[0], -1)
            else:
                # one process (no buckets): a plain, persistent p.grad -- the optimizer graph captured its address
                store = self.__dict__.setdefault("_dense_factor_grads", {})
                if id(W) not in store:
                    store[id(W)] = torch.empty(W.shape[0], W.numel() // W.shape[0], dtype=torch.float32, device=W.device)
                out = store[id(W)]
                W.grad = out.view(W.shape)
            K = out.shape[1]
            ops.gemm(L.GEMM_TN, gdy, 0, W.shape[0], gA, 0, K, out, K, W.shape[0], K, rows, compute=compute)

    def close(self):
        """Give the optimizer back: it reads p.grad again (not this step's 16-bit bucket views) and holds no gradient factors of
        this step's graphs.  Call before running train_step / GraphedTrainStep on the same optimizer (both also reset it)."""
        if hasattr(self.opt, "set_external_grads"):
            self.opt.set_external_grads(None)
        if hasattr(self.opt, "set_factored"):
            self.opt.set_factored(None)

    def _bucket16_back(self):
        """16-bit buckets with an optimizer that reads p.grad (stock torch optimizers; the gloo CPU tests): the averaged 16-bit
        values go back into the fp32 bucket views.  FusedAdamW reads the 16-bit buckets itself (set_external_grads): p.grad then
        keeps this rank's LOCAL fp32 gradients and averaged_grads() returns what the optimizer consumes."""
        if self.dist and self.bucket16 and not hasattr(self.opt, "set_external_grads"):
            for k in range(len(self.buckets)):
                self.flat[k][0].copy_(self.flat16[k][0])

    def averaged_grads(self):
        """{parameter: the averaged gradient tensor the optimizer reads} of the last step (views of the buckets)."""
        src = self.flat16 if (self.bucket16 and hasattr(self.opt, "set_external_grads")) else self.flat
        return {p: v for (_, views), ps in zip(src, self.buckets) for p, v in zip(ps, views)}

    def _from_early(self, t, cut2):
        """Is head-boundary tensor t produced in front of the inner cut (the features tapped off the early stages)?  Its
        autograd history then reaches an early-trunk parameter; the late trunk's outputs end at the cut's stand-in, the audio
        trunk's at its own parameters."""
        if not cut2:
            return False
        early = {id(p) for p in self.early_params}
        seen, stack = set(), [t.grad_fn]      # `seen` holds the node objects themselves: ids of dead wrappers get re-used
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            v = getattr(fn, "variable", None)             # AccumulateGrad node of a leaf
            if v is not None and id(v) in early:
                return True
            stack.extend(f for f, _ in fn.next_functions)
        return False

    # ------------------------------------------------------------------ replay
    def run(self, batch=None, lr: Optional[float] = None, timed: bool = False):
        if batch is not None:
            for k in self.static:
                if batch[k] is not self.static[k]:
                    self.static[k].copy_(batch[k], non_blocking=True)
        if lr is not None:
            set_lr(self.opt, lr)
        if hasattr(self.core, "_refresh_w16"):
            self.core._refresh_w16()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if timed else None
        host = [] if os.environ.get("CSTS_CHAIN_HOST_TIMES") else None          # diagnostics: host time of every stage of this call (us)
        def mark(i):
            if timed:
                ev[i].record()
            if host is not None:
                host.append(time.perf_counter())
        mark(0)
        self.graphs["fwd"].replay()
        mark(1)
        if "loss" in self.graphs:
            self._loss_gather(self.dist)
            self.graphs["loss"].replay()
            self._loss_scatter(self.dist)
            res = self.result
        else:
            res = self._loss_eager()
        mark(2)
        self.graphs["bwd_head"].replay()
        works = [self._all_reduce(0)] if self.dist else []
        works += self._exchange_factors(self.dist)
        mark(3)
        self.graphs["bwd_trunk"].replay()
        if self.dist:
            works.append(self._all_reduce(1))
        if self.trunk_cut:
            self.graphs["bwd_trunk_early"].replay()
            if self.dist:
                works.append(self._all_reduce(2))
        for w in works:
            w.wait()
        self._bucket16_back()
        self._finish_factors()
        mark(4)
        self.graphs["opt"].replay()
        mark(5)
        if timed:
            self.events = ev
        if host is not None:
            self.host_us = [round((b - a) * 1e6) for a, b in zip(host, host[1:])]     # fwd launch | loss section | head launch + collectives | trunk launches + waits + factor finish | opt launch
        return res

    def segment_ms(self):
        """(forward, loss, backward head, backward trunks [both parts + exposed all-reduce], optimizer) of the last
        run(timed=True)."""
        torch.cuda.synchronize()
        e = self.events
        return [e[i].elapsed_time(e[i + 1]) for i in range(5)]
