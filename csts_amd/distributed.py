"""Collectives of the training step (slowfast/utils/distributed.py:15-90 and the implicit DDP all-reduce of
slowfast/models/build.py:44-46), MI355X-first: one process per GPU, torch.distributed backend "nccl" (= RCCL
over xGMI), gradient buckets reduced in gradient-READY order on RCCL's own stream while backward continues.

The EgoNCE embedding gather has a correct backward here: the reference's AllGather_multi.backward slices with
a hard-coded rank 0 (distributed.py:23,28-32) and silently mis-routes gradients on ranks != 0; here every rank
takes the rows of the all-reduced gradient that belong to it."""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops as _ops


_FORCE = False    # tests: exercise the collective code paths on a 1-rank process group
_LOCAL = [0]      # > 0: inside local_only()


def is_dist() -> bool:
    return _LOCAL[0] == 0 and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


class local_only:
    """Context manager: the code inside runs on THIS rank alone (a check that only rank 0 performs, an eval pass): the
    collective wrappers of this module behave as in a single process instead of waiting for ranks that never come."""

    def __enter__(self):
        _LOCAL[0] += 1

    def __exit__(self, *a):
        _LOCAL[0] -= 1


class _AllGatherWithGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tensor):
        world = dist.get_world_size()
        out = [torch.empty_like(tensor) for _ in range(world)]
        dist.all_gather(out, tensor.contiguous())
        ctx.rank = dist.get_rank()
        ctx.bs = tensor.shape[0]
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, grad_output):
        # every rank computed the loss on the full gathered batch: d(sum of per-rank losses)/d(local rows) is the SUM
        # over ranks of that rank's grad rows; DDP-style averaging of the loss is applied by the caller's grad mean.
        g = grad_output.contiguous()
        dist.all_reduce(g)
        return g[ctx.bs * ctx.rank: ctx.bs * (ctx.rank + 1)]


def all_gather_with_grad(tensors: List[torch.Tensor]) -> List[torch.Tensor]:
    """distributed.py:35-49 (used for the EgoNCE embeddings, train_avgaze_net.py:82-83)."""
    if not is_dist():
        return list(tensors)
    return [_AllGatherWithGrad.apply(t) for t in tensors]


def all_gather(tensors):
    """distributed.py:52-71."""
    if not is_dist():
        return list(tensors)
    world = dist.get_world_size()
    out = []
    for t in tensors:
        buf = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(buf, t.contiguous())
        out.append(torch.cat(buf, dim=0))
    return out


def all_reduce(tensors, average=True):
    """distributed.py:74-90.  All scalars go out as ONE fused collective."""
    if not is_dist():
        return tensors
    flat = torch.stack([t.detach().reshape(()).float() for t in tensors])
    dist.all_reduce(flat)
    if average:
        flat /= dist.get_world_size()
    return [flat[i] for i in range(len(tensors))]


class GradAllReduce(nn.Module):
    """Data-parallel wrapper: bucketed mean all-reduce of gradients overlapped with backward.

    Buckets are filled in the order gradients become READY (reverse registration order is used to assign
    parameters to buckets once, after the first backward the observed order is kept), flattened into one
    contiguous fp32 buffer per bucket and reduced asynchronously; ``finish()`` waits and scatters the means back.
    The three 37.7 M-parameter fusion convs (60 % of all gradient bytes) become ready right after the decoder
    backward, so their buckets travel over xGMI underneath the whole trunk backward."""

    def __init__(self, module: nn.Module, bucket_mb: int = 64):
        super().__init__()
        self.module = module
        self.hooks_enabled = True       # csts_amd.train.SegmentedTrainStep drives the buckets itself and switches this off
        self.world = dist.get_world_size() if is_dist() else 1
        self.bucket_bytes = int(bucket_mb) * (1 << 20)
        self._late = []
        late_ids = set()
        if self.world > 1 or _FORCE:
            # Bucket hooks read gradients in the middle of backward.  Linear weight gradients therefore run in line (no
            # grouped end-of-backward launch).  The LayerNorm / pooling-stencil parameters, whose second-stage
            # reductions ARE deferred to the end of backward (ops.flush_deferred), are kept out of the hook-driven
            # buckets and reduced in one small bucket by finish() (they total ~0.1 % of the gradient bytes).
            _ops.GROUP_WGRADS = "never"
            for name, p in module.named_parameters():
                if p.requires_grad and any(t in name for t in (".norm1.", ".norm2.", ".norm_q.", ".norm_k.", ".norm_v.",
                                                               ".pool_q.", ".pool_k.", ".pool_v.", ".upsample_q.")):
                    self._late.append(p)
                    late_ids.add(id(p))
        self._params = [p for p in module.parameters() if p.requires_grad and id(p) not in late_ids]
        self._assign(list(reversed(self._params)))
        self._ready_order: List[torch.nn.Parameter] = []
        self._observed = False
        self._pending = []
        self._avg_op = None
        if is_dist() and dist.get_backend() == "nccl":
            self._avg_op = dist.ReduceOp.AVG
        for p in self._params:
            p.register_post_accumulate_grad_hook(self._hook)
        if is_dist():   # replicas start identical
            for p in module.parameters():
                dist.broadcast(p.data, src=0)

    def _assign(self, ordered):
        self._buckets, self._bucket_of = [], {}
        cur, cur_bytes = [], 0
        for p in ordered:
            nb = p.numel() * 4
            if cur and cur_bytes + nb > self.bucket_bytes:
                self._buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._buckets.append(cur)
        for bi, b in enumerate(self._buckets):
            for p in b:
                self._bucket_of[p] = bi
                p._csts_bucket = bi          # read by the per-gradient hook (524 calls per step: keep it to a few lookups)
        self._count = [0] * len(self._buckets)
        self._bucket_len = [len(b) for b in self._buckets]
        self._bucket_streams = {}
        self._iters_since_assign = 0

    def _hook(self, p):
        if not self.hooks_enabled:
            return
        if not self._observed:
            self._ready_order.append(p)
        bi = p._csts_bucket
        c = self._count[bi] + 1
        self._count[bi] = c
        if self._iters_since_assign < 2 and (self.world > 1 or _FORCE):
            # backward replays on more than one HIP stream (the audio trunk has its own): learn, during the first two
            # iterations after a bucket assignment, which streams produce gradients of this bucket (the graph is static),
            # so that its launch can wait for them
            cur = self._cur_stream(p.grad)
            if cur is not None:
                self._bucket_streams.setdefault(bi, {})[cur.cuda_stream] = cur
        if c == self._bucket_len[bi]:
            self._launch(bi)

    def _cur_stream(self, grad):
        """The HIP stream the calling autograd node runs on (None for CPU tensors; tests substitute a stub)."""
        return torch.cuda.current_stream() if grad.is_cuda else None

    def _launch(self, bi):
        if self.world == 1 and not _FORCE:
            return
        ps = self._buckets[bi]
        cur = self._cur_stream(ps[0].grad)
        if cur is not None:
            for sid, st in self._bucket_streams.get(bi, {}).items():
                if sid != cur.cuda_stream:
                    cur.wait_stream(st)      # gradients of this bucket that were produced on another stream
        # one flat fp32 buffer per bucket, every tensor's slot 16-byte aligned (the optimizer kernels read the averaged
        # gradients straight out of it through p.grad views); filled by ONE multi-tensor copy
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        flat = torch.empty(total, dtype=torch.float32, device=ps[0].grad.device)
        views = [flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)]
        torch._foreach_copy_(views, [p.grad for p in ps])
        if self._avg_op is not None:
            work = dist.all_reduce(flat, op=self._avg_op, async_op=True)      # RCCL averages in the collective itself
        else:
            flat /= self.world
            work = dist.all_reduce(flat, async_op=True)
        self._pending.append((work, flat, ps, views))

    def finish(self):
        """Wait for the outstanding buckets and hand the averaged gradients back (call after backward): every p.grad
        becomes a VIEW of its bucket's flat buffer -- no copy back."""
        if not self.hooks_enabled:
            return
        if self._late and (self.world > 1 or _FORCE):
            # complete by now: the deferred reductions ran in the autograd final callback, on this stream
            ps = [p for p in self._late if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
            if self._avg_op is not None:
                dist.all_reduce(flat, op=self._avg_op)
            else:
                flat /= self.world
                dist.all_reduce(flat)
            off = 0
            for p in ps:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        for work, flat, ps, views in self._pending:
            work.wait()
            if flat.is_cuda:
                flat.record_stream(torch.cuda.current_stream())    # allocated on the hook's stream, read from here on
            for p, v in zip(ps, views):
                p.grad = v
        self._pending = []
        self._count = [0] * len(self._buckets)
        self._iters_since_assign += 1
        if not self._observed and self._ready_order:
            self._observed = True
            if len(self._ready_order) == len(self._params):
                self._assign(self._ready_order)     # keep the measured gradient-ready order from now on

    def forward(self, *a, **k):
        return self.module(*a, **k)
