"""torch.autograd Functions over the libcsts_hip.so C ABI.

PyTorch is plumbing here (device memory from the caching allocator, the current HIP stream, the
autograd tape); every FLOP of the CSTS path runs in the hand-written gfx950 kernels.  There is no
eager fallback: tensors must be on a GPU and the library must load, otherwise these raise.
Each Function cites the reference op it replaces (paths under the reference repo).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import weakref
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import lib as L

F32, BF16 = L.F32, L.BF16


def _lib():
    return L.load()


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == L.half_dtype():       # bfloat16, or float16 in a process that runs the fp16 build (lib.set_half)
        return BF16
    raise L.CstsError(f"unsupported dtype {t.dtype} (this process runs the {L.HALF} kernel library)")


def torch_dtype(dt: int):
    return torch.float32 if dt == F32 else L.half_dtype()


try:
    _raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:          # CPU-only build of torch: ops raise before they get here
    _raw_stream = _cur_dev = None


def _stream():
    """Raw hipStream_t of torch's current stream (torch.cuda.current_stream().cuda_stream costs ~4 us per call, and
    there are ~1200 launches per step)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], off_elems: int = 0):
    if t is None:
        return None
    return t.data_ptr() + off_elems * t.element_size()


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.CstsError("csts_amd ops need GPU tensors (no CPU fallback); got a CPU tensor")


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


_host_tables = []       # weak references to every HostTable of the process


def refill_capture_pools():
    """Called by the graph-capturing steps (csts_amd.train) right before a capture pass: every table gets its full set of
    capture buffers back, so a process may capture any number of steps one after the other."""
    for r in list(_host_tables):
        t = r()
        if t is None:
            _host_tables.remove(r)
        else:
            t.refill()


class HostTable:
    """Small host -> device table upload that is safe while the host runs ahead of the GPU (a ring of pinned buffers,
    a slot is reused only after its copy has executed) and under HIP-graph capture (pre-allocated pinned buffers that
    are never written again, so a replay copies the captured contents)."""

    def __init__(self, nbytes: int, device, ring: int = 4, captures: int = 8):
        self.nbytes, self.device = nbytes, device
        self._ring = [torch.zeros(nbytes, dtype=torch.uint8).pin_memory() for _ in range(ring)]
        self._ev = [None] * ring
        self._pos = 0
        self._captures = captures
        self._pool = [torch.zeros(nbytes, dtype=torch.uint8).pin_memory() for _ in range(captures)]
        self._captured = []
        self.dev = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _host_tables.append(weakref.ref(self))

    def refill(self):
        """Top the capture pool up again (outside a capture: pinned allocations are not allowed while one is open)."""
        while len(self._pool) < self._captures:
            self._pool.append(torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory())

    def upload(self, payload: bytes) -> int:
        n = len(payload)
        if n > self.nbytes:
            raise L.CstsError(f"HostTable: {n} bytes > capacity {self.nbytes}")
        src = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        if torch.cuda.is_current_stream_capturing():
            if not self._pool:
                raise L.CstsError("HostTable: out of capture buffers")
            host = self._pool.pop()
            host[:n].copy_(src)
            self._captured.append(host)
            self.dev[:n].copy_(host[:n], non_blocking=True)
        else:
            k = self._pos
            self._pos = (k + 1) % len(self._ring)
            if self._ev[k] is not None:
                self._ev[k].synchronize()
            self._ring[k][:n].copy_(src)
            self.dev[:n].copy_(self._ring[k][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._ev[k] = ev
        return self.dev.data_ptr()


# Second stages of the cross-workgroup reductions of backward (LayerNorm dgamma/dbeta, stencil dweight): ~190 tiny
# launches per step when done in line.  Inside a backward pass they are deferred instead: the first-stage kernels leave
# their partial rows in a workspace, and ONE csts_reduce_rows_batched launch finishes all of them from an autograd final
# callback (which runs on the caller's stream after the engine has joined every stream backward used).
#
# A deferred gradient is NOT handed to autograd (an unfilled buffer would be adopted by AccumulateGrad only while the
# parameter has no .grad yet, and be ADDED, garbage and all, when it has one: gradient accumulation, a module applied twice).
# The Function returns None for that input; the final callback launches the finishing kernels and then hands every
# finished tensor to its leaf parameter itself: p.grad = g, or p.grad += g when a gradient is already there.  Strong
# references to operands, partial rows and results are held until then.
#
# The queues belong to ONE backward pass, identified by autograd's graph-task id: final callbacks are dropped when
# backward raises, so a pass that finds another pass's leftovers discards them (stale device addresses) and registers
# its own callback.
_deferred = []          # (ws tensor, out tensor, nrows, ncols)
_assign = []            # (leaf parameter, finished gradient tensor)
_deferred_task = [-1]   # graph-task id the queues belong to (-1: none)
_deferred_tables = {}   # device index -> HostTable
DEFER_REDUCTIONS = os.environ.get("CSTS_DEFER_REDUCE", "1") != "0"


def reset_deferred():
    """Drop everything queued for an end-of-backward flush (used when a backward pass died before its final callback)."""
    if _wg_side_used[0]:                # a dead pass may have launches in flight that read / write what is dropped below
        for st in _wg_side.values():
            st.synchronize()
        _wg_side_used[0] = False
    if _sw_used[0]:
        for st in _sw_side.values():
            st.synchronize()
        _sw_used[0] = False
    _sw_keep.clear()
    _swq.clear()
    _swq_count[0] = 0
    _sw_prod.clear()
    _swq_early_keep.clear()
    _swq_early_used[0] = False
    _deferred.clear()
    _assign.clear()
    _wgq.clear()
    _wg_pending.clear()
    _wg_prod.clear()
    _wg_work[0] = 0.0
    _w8_count[0] = 0
    _deferred_task[0] = -1


def _can_defer(*params) -> bool:
    """True when the caller may leave its gradient for the end-of-backward flush: inside a backward pass, and every
    given parameter is a leaf the flush can assign to (None entries are ignored)."""
    if not DEFER_REDUCTIONS:
        return False
    for p_ in params:
        if p_ is not None and not (p_.is_leaf and p_.requires_grad):
            return False
    tid = torch._C._current_graph_task_id()
    if tid < 0:                        # not inside a backward pass
        return False
    if _deferred_task[0] != tid:
        reset_deferred()               # leftovers of a pass that raised before its callback ran
        torch.autograd.Variable._execution_engine.queue_callback(flush_deferred)
        _deferred_task[0] = tid
    return True


def _defer(ws: torch.Tensor, out: torch.Tensor, nrows: int, ncols: int):
    _deferred.append((ws, out, int(nrows), int(ncols)))


# Gradient targets (csts_amd.train.SegmentedTrainStep, data-parallel chain): parameter storage address -> a view of a flat
# all-reduce bucket shaped like the parameter.  A producer that allocates a parameter gradient (queued Linear weight gradients,
# the fusion convs' TN GEMM) writes it THERE, so the bucket needs no copy afterwards.  One use per backward pass: a second
# gradient for the same parameter (a module applied twice) gets an ordinary buffer and is accumulated as before.
_grad_targets = {}
_grad_targets_used = set()


def set_grad_targets(pairs=None):
    _grad_targets.clear()
    _grad_targets_used.clear()
    for p_, v_ in (pairs or []):
        if v_.dtype == torch.float32 and v_.is_contiguous() and v_.numel() == p_.numel():
            _grad_targets[p_.data_ptr()] = v_


def _grad_buffer(param_like, shape, device):
    """fp32 buffer for the gradient of `param_like` (a Parameter or a saved alias of it): its bucket view if one is registered,
    unused in this backward pass and the parameter has no gradient yet; else a fresh tensor."""
    key = param_like.data_ptr()
    v = _grad_targets.get(key)
    if v is not None and key not in _grad_targets_used and getattr(param_like, "grad", None) is None and v.device == device:
        _grad_targets_used.add(key)
        return v.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=device)


def _assign_later(param, grad):
    if param is not None and grad is not None:
        _assign.append((param, grad))


def _hand_over():
    """Give the finished gradients to their parameters (stream-ordered after the finishing kernels)."""
    for p_, g_ in _assign:
        if p_.grad is None:
            p_.grad = g_ if g_.dtype == p_.dtype else g_.to(p_.dtype)
        else:
            p_.grad.add_(g_)
    _assign.clear()
    _grad_targets_used.clear()


def _targets_reset():
    _grad_targets_used.clear()


def flush_deferred():
    """Finish the queued weight gradients and every deferred reduction on the current stream, then hand the results to
    their parameters (idempotent)."""
    _deferred_task[0] = -1
    if _w8_count[0]:
        _w8_total[0] = _w8_count[0]
    _w8_count[0] = 0
    # the grouped stencil weight gradients (0.8 ms of load-latency-bound work, no matrix math) run on a side stream BESIDE the
    # grouped Linear weight gradients (MFMA-bound, one persistent workgroup per CU whose last items leave most CUs idle)
    _swq_count[0] = 0
    _sw_prod.clear()
    tail = None
    if _swq and _wgq and STENCIL_TAIL_SIDE and _swq[0][5][0].is_cuda:
        cur = torch.cuda.current_stream()
        tail = _tail_side.get(cur.device.index)
        if tail is None:
            tail = _tail_side[cur.device.index] = torch.cuda.Stream(device=cur.device)
        tail.wait_stream(cur)
        if W8_PARALLEL:
            # Round 5 experiment: the 192 x 384 class (310 items on 256 CUs: its second round fills a fifth of the chip) FIRST and alone on this
            # stream, every other weight-gradient launch on the side stream -- they can take the CUs its second round leaves idle
            flush_wgrads(w8="only")
        with torch.cuda.stream(tail):
            keep = flush_stencil_wgrads()
            if W8_PARALLEL:
                flush_wgrads(w8="exclude")
    flush_wgrads()
    if tail is not None:
        torch.cuda.current_stream().wait_stream(tail)
        del keep        # allocated on this stream, read on `tail`: alive until the join above
    else:
        flush_stencil_wgrads()
    if _swq_early_used[0]:
        for st in _tail_side.values():
            torch.cuda.current_stream().wait_stream(st)
        _swq_early_used[0] = False
        _swq_early_keep.clear()
    _join_wgrad_side()
    _join_stencil_side()
    if not _deferred:
        _hand_over()
        return
    # wide & shallow (split-K slabs) after narrow & deep (LayerNorm / stencil partial rows): two kernels
    wide = lambda it: it[3] >= 8192 and it[3] % 4 == 0 and it[2] <= 64
    items = sorted(_deferred, key=lambda it: (wide(it), it[3]))
    _deferred.clear()
    dev = items[0][0].device
    tab = _deferred_tables.get(dev.index)
    if tab is None:
        tab = _deferred_tables[dev.index] = HostTable(C.sizeof(L.ReduceDesc) * 2048, dev, ring=4, captures=16)
    # one launch per size class (the grid is sized by the widest reduction of the launch)
    descs = (L.ReduceDesc * len(items))()
    for i, (ws, out, nrows, ncols) in enumerate(items):
        descs[i].ws, descs[i].out = ws.data_ptr(), out.data_ptr()
        descs[i].nrows, descs[i].ncols, descs[i].scale = nrows, ncols, 1.0
    raw = bytes(descs)
    sz = C.sizeof(L.ReduceDesc)
    if len(items) > 2048:
        raise L.CstsError("too many deferred reductions in one backward pass")
    base = tab.upload(raw)
    lo = 0
    while lo < len(items):
        hi = lo + 1
        w = wide(items[lo])
        while hi < len(items) and wide(items[hi]) == w and items[hi][3] <= (64 if w else 4) * items[lo][3]:
            hi += 1
        fn = _lib().csts_reduce_rows_wide if w else _lib().csts_reduce_rows_batched
        L.check(fn(base + lo * sz, hi - lo, items[hi - 1][3], _stream()), "csts_reduce_rows_batched")
        lo = hi
    del items
    _hand_over()


# Weight gradients of the Linear layers (dW = dY^T X) are off the critical path of backward: inside a backward pass they
# are queued and computed by csts_wgrad_grouped at the end, all layers in one launch per dY dtype.  A weight gradient on
# its own needs a deep split-K to fill the chip; together they provide thousands of (tile, token-chunk) work items, so
# chunks are long (WGRAD_CHUNK tokens) and most layers need no split-K partials at all.
# "capture" (default): only while the step is being captured into a HIP graph -- replayed, the grouped tail costs nothing
# on the host and the graph's private pool makes the longer tensor lifetimes free.  In eager mode the end-of-backward
# release of every queued operand at once fragments torch's caching allocator (hipMalloc stalls of ~20 ms in the next
# forward were measured), so eager steps keep the in-line split-K weight gradients unless CSTS_GROUP_WGRADS=1 forces
# grouping; CSTS_GROUP_WGRADS=0 switches it off everywhere.
GROUP_WGRADS = {"0": "never", "1": "always"}.get(os.environ.get("CSTS_GROUP_WGRADS", ""), "capture")
# 192 x 384 tiles on 8-wave workgroups (wgrad8.hip: half the operand bytes per FLOP, k-tiles by LDS-DMA into a 2-stage ring,
# no ds_write phase) for the layers they divide (the 384- and 768-channel stages).  Measured on MI355X
# (profiles/r2_wgrad8_ab.txt): the register-staged first version 1541 us against the 1497-1565 us those layers cost as
# 128 x 128 / 256 x 128 items (neutral); the LDS-DMA version 1472 us (-6 %), 24.38-24.42 vs 24.37-24.48 ms per step over three
# interleaved pairs.  On by default; CSTS_WGRAD8=0 is the A/B switch.  Token chunk: 8192 (4096 / 2048 measured +0.05 / +0.15 ms).
WGRAD8 = os.environ.get("CSTS_WGRAD8", "1") != "0"
WGRAD8_CHUNK = int(os.environ.get("CSTS_WGRAD8_CHUNK", "8192"))
# Round 5: the thin layers (output and input features multiples of 96 that the 192 x 384 class does not take; bf16 dY) as 96 x 96 tiles, one
# (tile, token chunk) item per WAVE of csts_wgrad_grouped5 (wgrad5.hip: every wave its own LDS-DMA stream, no workgroup barrier)
WG_DUMP = os.environ.get("CSTS_WGRAD_DUMP", "")
WGRAD_CAST_F32 = os.environ.get("CSTS_WGRAD_CAST_F32", "0") == "1"      # measured neutral (19.39 vs 19.39 ms, gpurun_out/r5al): off
WGRAD5 = os.environ.get("CSTS_WGRAD5", "1") != "0"
WGRAD5_CHUNK = int(os.environ.get("CSTS_WGRAD5_CHUNK", "4096"))
WGRAD5_STRIDED = os.environ.get("CSTS_WGRAD5_STRIDED", "1") != "0"
WGRAD5_MIN = int(os.environ.get("CSTS_WGRAD5_MIN", "96"))          # layers with min(N, K) below this stay on the 128-wide classes
# first: beside the stencil weight gradients; mid (default): after the 128-wide classes -- the stencil launch then starts beside the fp32-dY class
# and the remaining 128-wide items (wgrad5 and the stencil kernel side by side take the SUM of their times: profiles/r5_wgrad5.txt)
WGRAD5_POS = os.environ.get("CSTS_WGRAD5_POS", "mid")
WGRAD_CHUNK = int(os.environ.get("CSTS_WGRAD_CHUNK", "8192"))   # tokens per work item (measured per step: 4096 -> 24.93 ms, 8192 -> 24.95, 16384 -> 25.47)
_wgq = []               # (dY, X, dW, db, tokens, N_out, K_in)
WG_STATS = None         # a list while bench.py instruments a step: (C-ABI entry, algorithmic bytes, flop) per grouped launch
_wg_tables = {}
# Optional: every WG_FLUSH_GFLOP of queued work goes out as its own grouped launch on a SIDE stream, beside the rest of
# backward (the idea: backward's kernels are small, and the audio trunk on its own stream was worth 8 ms of a 41 ms step).
# Operands, slabs and results stay referenced until the final callback has joined the side stream.
# MEASURED SLOWER on MI355X (bench.py, same box, ms per step): one launch at the end 25.19-25.28, flushes of 600 / 350 / 200
# GFLOP on the side stream 25.62 / 25.46-25.51 / 26.74 -- backward does not leave the matrix pipes idle enough for a second
# MFMA-heavy kernel beside it.  Default 0 (off); the switch stays for re-measuring on other workloads.
WG_FLUSH_FLOP = float(os.environ.get("CSTS_WGRAD_FLUSH_GFLOP", "0")) * 1e9
_wg_work = [0.0]
_wg_prod = {}           # raw stream id -> torch stream that produced operands queued since the last flush
_wg_side = {}           # device index -> side stream
_wg_pending = []        # strong references of side-stream flushes in flight
_wg_side_used = [False]


def queue_wgrad(dY, X, tokens, N, K, Wp, bp):
    """Queue dW[N,K] = dY[tokens,N]^T X[tokens,K] (+ db[N] when bp is given) for the grouped launch at the end of this
    backward pass, which also hands the results to the leaf parameters Wp / bp (see the note on deferred gradients
    above).  Returns False when the problem has to run in line."""
    if GROUP_WGRADS == "never" or (GROUP_WGRADS == "capture" and not torch.cuda.is_current_stream_capturing()):
        return False
    if not (DEFER_REDUCTIONS and X.dtype == L.half_dtype() and dY.dtype in (torch.float32, L.half_dtype())
            and N % 8 == 0 and K % 8 == 0 and dY.is_contiguous() and X.is_contiguous() and tokens >= 256):
        return False
    if Wp is None or Wp.dtype != torch.float32 or tuple(Wp.shape) != (N, K) or not _can_defer(Wp, bp):
        return False
    dW = _grad_buffer(Wp, (N, K), dY.device)
    db = _grad_buffer(bp, (N,), dY.device) if bp is not None else None
    _wgq.append((dY, X, dW, db, tokens, N, K))
    _assign_later(Wp, dW)
    _assign_later(bp, db)
    cur = torch.cuda.current_stream()
    _wg_prod[cur.cuda_stream] = cur
    _wg_work[0] += 2.0 * tokens * N * K
    if WG_FLUSH_FLOP > 0 and _wg_work[0] >= WG_FLUSH_FLOP:
        flush_wgrads(side=True)
    if W8_EARLY_WGS > 0 and _is_w8(dY, tokens, N, K):
        # the 192 x 384 class goes out EARLY, on few workgroups, as soon as its last problem of this backward pass is queued (the count
        # of the previous pass tells which one that is; a wrong guess only moves work between this launch and the final one)
        _w8_count[0] += 1
        if _w8_count[0] == _w8_total[0]:
            flush_wgrads(side=True, only_w8=True)
    return True


# Round 5 (CSTS_WGRAD8_EARLY_WGS = n > 0): the grouped weight gradients of the 384- / 768-channel stages (wgrad8: matrix-bound, 1.1 ms, one
# 144 KB-LDS workgroup per CU) are launched on n workgroups on the side stream as soon as the backward pass has produced their last operand
# -- the rest of backward (the 96- / 192-channel stages: large-M, memory-bound kernels, a third of the trunk backward) runs beside them on
# the remaining CUs -- instead of on all CUs after the pass (profiles/r5_wgrad8_early_ab.txt).
W8_EARLY_WGS = int(os.environ.get("CSTS_WGRAD8_EARLY_WGS", "0"))
_w8_count = [0]         # 192 x 384-class problems queued so far in this backward pass
_w8_total = [0]         # ... in the whole previous pass


def _is_w8(dY, tokens, N, K):
    return bool(WGRAD8 and dY.dtype != torch.float32 and N % 192 == 0 and K % 384 == 0 and tokens % 64 == 0 and WGRAD8_CHUNK % 64 == 0)


_WG_DTYPE = None
_wg_plans = {}          # (tuple of (tokens, N, K) per problem) -> cached work-item layout


def _wg_dtype():
    global _WG_DTYPE
    if _WG_DTYPE is None:
        import numpy as np
        _WG_DTYPE = np.dtype([("A", "u8"), ("B", "u8"), ("C", "u8"), ("colsum", "u8"), ("lda", "i8"), ("ldb", "i8"), ("ldc", "i8"),
                              ("kbeg", "i8"), ("kend", "i8"), ("M", "i4"), ("N", "i4"), ("m0", "i4"), ("n0", "i4")], align=True)
        assert _WG_DTYPE.itemsize == C.sizeof(L.WgradItem)
    return _WG_DTYPE


def _wg_plan(sig, rows=128, cols=128, chunk_tokens=None, strided=False):
    """Work-item layout of one grouped launch for the problem list `sig` = ((tokens, N, K), ...): everything except the
    device addresses, which change from step to step.  Cached: building ~7000 items in Python costs ~15 ms.

    Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2: the tiles of one
    (layer, token chunk) go to ONE XCD, these groups are spread over the XCDs by load (long chunks first), and XCD x's list
    is laid out at positions x, x + 8, x + 16, ... (rocprofv3: 7x the algorithmic HBM bytes when consecutive tiles landed
    on different XCDs).  Padding slots keep A == NULL (the kernel skips them)."""
    import numpy as np
    CH = chunk_tokens or WGRAD_CHUNK
    plan = _wg_plans.get((sig, rows, cols, CH, strided))
    if plan is not None:
        return plan
    # (Cutting the layers with the fewest tiles into half-length chunks, so that the 192 x 384 class's 310 full-length items
    # of the b=4 step do not leave 54 of them for a second, almost empty round on 256 CUs, was measured: kernel 1335 ->
    # 1276 us, step unchanged -- the partial slabs and their reduction cost what the balance returns.)
    CHs = [CH] * len(sig)
    groups = []
    for pi, (tokens, N, K) in enumerate(sig):
        CHp = CHs[pi]
        nch = -(-tokens // CHp)
        for c in range(nch):
            kb, ke = c * CHp, min(tokens, (c + 1) * CHp)
            if strided:
                # csts_wgrad_grouped5: the nch items of a tile INTERLEAVE 16-token stages (item c takes stages c, c + nch, ...: kbeg = its
                # first stage, kend = the layer's tokens, M = the stage stride) -- the waves that work on one layer then read ONE moving
                # window of its token rows instead of nch distant ranges (2048 wave streams of 3 KB bursts: 3.0 TB/s, tools/wgrad5_bench.py)
                nst = -(-tokens // 16)
                groups.append([(16 * len(range(c, nst, nch)), pi, c, 16 * c, tokens, m0, n0, nch) for m0 in range(0, N, rows) for n0 in range(0, K, cols)])
                continue
            # one group = ALL tiles of this (layer, token chunk): they run together on one XCD and stream the same
            # token range in near lockstep, so its L2 serves every dY / X panel slice to all the tiles that share it
            # (grouping by tile-row only reused the dY panel: rocprofv3 still counted 18.6 GB per launch)
            groups.append([(ke - kb, pi, c, kb, ke, m0, n0, 0) for m0 in range(0, N, rows) for n0 in range(0, K, cols)])
    # two levels of longest-first: groups go to the least-loaded XCD in order of their total work; inside an XCD's list the
    # groups with the longest ITEMS start first (a 8192-token item runs 4 x as long as a 2048-token one: started last it
    # is the tail of the launch, with most CUs idle -- 1448 -> 1317 us for the 192 x 384 class of the b=4 step)
    groups.sort(key=lambda g_: -g_[0][0] * len(g_))
    glists, load = [[] for _ in range(8)], [0] * 8
    for g_ in groups:
        x = load.index(min(load))
        glists[x].append(g_)
        load[x] += g_[0][0] * len(g_)
    lists = []
    for gl in glists:
        gl.sort(key=lambda g_: -g_[0][0])          # stable: equal item lengths keep the by-work order
        lists.append([it for g_ in gl for it in g_])
    depth = max(len(l_) for l_ in lists)
    n_items = depth * 8
    if n_items > 16384:
        raise L.CstsError("too many grouped weight-gradient work items")
    tmpl = np.zeros(n_items, dtype=_wg_dtype())
    pidx = np.full(n_items, -1, dtype=np.int64)
    chunk = np.zeros(n_items, dtype=np.int64)
    for x in range(8):
        for slot, (_, pi, c, kb, ke, m0, n0, stride) in enumerate(lists[x]):
            i = slot * 8 + x
            tokens, N, K = sig[pi]
            tmpl[i] = (0, 0, 0, 0, N, K, K, kb, ke, stride if rows == 96 else N, K, m0, n0)     # 96 x 96 per-wave class: M carries the stage step
            pidx[i], chunk[i] = pi, c
    valid = pidx >= 0
    plan = _wg_plans[(sig, rows, cols, CH, strided)] = (tmpl, valid, pidx[valid], chunk[valid], n_items, CHs)
    return plan


def flush_wgrads(side: bool = False, only_w8: bool = False, w8: str = "all"):
    """Launch the queued weight gradients: on the current stream (end of backward), or -- side=True, from queue_wgrad in
    the middle of backward -- on the side stream, after everything the producing streams have enqueued so far.  only_w8: just the
    192 x 384-class problems (on W8_EARLY_WGS workgroups); everything else stays queued."""
    if not _wgq:
        return
    import numpy as np
    if only_w8 or w8 != "all":      # w8 = "only" / "exclude": the 192 x 384 class alone (a full launch) / everything but it; the rest stays queued
        q = [t for t in _wgq if _is_w8(t[0], t[4], t[5], t[6])]
        rest = [t for t in _wgq if not _is_w8(t[0], t[4], t[5], t[6])]
        if w8 == "exclude":
            q, rest = rest, q
        _wgq.clear()
        _wgq.extend(rest)
        if not q:
            return
    else:
        q = list(_wgq)
        _wgq.clear()
        _wg_work[0] = 0.0
    if WG_DUMP and not os.path.exists(WG_DUMP):      # diagnostics: the problem list of one flush, for tools/wgrad5_bench.py
        with open(WG_DUMP, "w") as f:
            for t in q:
                f.write(f"{'f32' if t[0].dtype == torch.float32 else 'h16'} {t[4]} {t[5]} {t[6]} {t[0].stride(0)} {t[1].stride(0)} {int(t[3] is not None)}\n")
    dev = q[0][0].device
    # (w8 = "exclude" runs on the tail side stream CONCURRENTLY with the "only" launch of the main stream: its own device table -- one table
    # for both let the second upload overwrite the items the first launch was still reading: a GPU memory fault, round 5)
    key = (dev.index, "x" if w8 == "exclude" else side)
    tab = _wg_tables.get(key)
    if tab is None:
        tab = _wg_tables[key] = HostTable(C.sizeof(L.WgradItem) * 24576, dev, ring=8, captures=40 if side else 16)
    keep = []
    launch_stream = None
    if side:
        launch_stream = _wg_side.get(dev.index)
        if launch_stream is None:
            # CSTS_WG_SIDE_PRIO=low: the early weight-gradient flushes on a LOW-priority stream (experiment, round 5: they should only
            # fill the CUs the activation-gradient chain leaves idle, not delay it)
            prio = 0
            if os.environ.get("CSTS_WG_SIDE_PRIO", "") == "low":
                try:
                    prio = max(torch.cuda.Stream.priority_range())
                except Exception:
                    prio = 0
            launch_stream = _wg_side[dev.index] = torch.cuda.Stream(device=dev, priority=prio)
        for st in _wg_prod.values():          # autograd runs every node of this device on ONE thread: whatever produced the
            ev = torch.cuda.Event()           # queued operands is already enqueued on these streams
            ev.record(st)
            launch_stream.wait_event(ev)
        _wg_side_used[0] = True
    _wg_prod.clear()
    # Round 5: fp32 dY of layers the LDS-DMA classes can take (a handful per step: the stage-transition projections, whose output gradient
    # has no 16-bit copy) is cast to the 16-bit type first -- the 128-wide fp32-dY class rounds it the same way while staging, but as 4
    # problems / ~330 workgroups of its own it took 0.28 ms per step for 153 MB (profiles/r5_final_mfma_util.txt); as 16-bit operands
    # they ride in the 192 x 384 / 96 x 96 launches.  Whole step: neutral -- the fp32 class runs beside the stencil weight gradients, which bound that
    # part of the tail either way -- so this is opt-in (CSTS_WGRAD_CAST_F32=1).
    if WGRAD_CAST_F32:
        with (torch.cuda.stream(launch_stream) if launch_stream is not None else contextlib.nullcontext()):
            for i, t in enumerate(q):
                if t[0].dtype == torch.float32 and (_is_w8(t[1], t[4], t[5], t[6]) or
                                                    (WGRAD5 and t[5] % 96 == 0 and t[6] % 96 == 0 and t[4] % 16 == 0 and min(t[5], t[6]) >= WGRAD5_MIN)):
                    d16 = torch.empty(t[0].shape, dtype=L.half_dtype(), device=t[0].device)
                    cast_into(d16, t[0])
                    q[i] = (d16,) + tuple(t[1:])
    # tile classes: 192 x 384 on 8-wave workgroups where it divides the layer (the 384- and 768-channel stages: half the
    # operand bytes per FLOP of a 128 x 128 tile); else 256 x 128 where 256 divides the output rows; else 128 x 128
    def tile_class(t):
        if t[0].dtype == torch.float32:
            return (True, 128)
        if WGRAD8 and t[5] % 192 == 0 and t[6] % 384 == 0 and t[4] % 64 == 0 and WGRAD8_CHUNK % 64 == 0:
            return (False, 192)
        if WGRAD5 and t[5] % 96 == 0 and t[6] % 96 == 0 and t[4] % 16 == 0 and WGRAD5_CHUNK % 16 == 0 and min(t[5], t[6]) >= WGRAD5_MIN:
            return (False, 96)
        return (False, 256 if t[5] % 256 == 0 else 128)
    pend = []           # (item table image, items, tile rows, fp32 dY) per tile class
    # WGRAD8_LAST (with CSTS_STENCIL_TAIL_SIDE=1): the 192 x 384 class -- one 144 KB-LDS workgroup per CU, nothing fits beside it -- goes
    # last, so that the grouped stencil weight gradients on the side stream (vector-bound, 20 KB of LDS, 124 registers) start beside the
    # 128-wide classes (memory-bound, 40 KB of LDS, 154 registers), which they CAN share a CU with
    order = (((False, 96), (False, 256), (False, 128), (True, 128), (False, 192)) if WGRAD8_LAST
             else ((False, 192), (False, 96), (False, 256), (False, 128), (True, 128)))
    if WGRAD8_LAST and WGRAD5_POS == "mid":
        order = ((False, 256), (False, 128), (True, 128), (False, 96), (False, 192))
    for a_f32, rows in order:
        probs = [t for t in q if tile_class(t) == (a_f32, rows)]
        if not probs:
            continue
        CH = WGRAD8_CHUNK if rows == 192 else (WGRAD5_CHUNK if rows == 96 else WGRAD_CHUNK)
        tmpl, valid, pidx, chunk, n_items, CHs = _wg_plan(tuple((t[4], t[5], t[6]) for t in probs), rows, {192: 384, 96: 96}.get(rows, 128), CH, strided=(rows == 96 and WGRAD5_STRIDED))
        A = np.empty(len(probs), dtype=np.uint64); B = np.empty_like(A); Cb = np.empty_like(A); Cs = np.zeros_like(A)
        cstride = np.zeros(len(probs), dtype=np.uint64); sstride = np.zeros_like(cstride)
        for i, (dY, X, dW, db, tokens, N, K) in enumerate(probs):
            A[i], B[i] = dY.data_ptr(), X.data_ptr()
            nch = -(-tokens // CHs[i])
            if nch == 1:
                Cb[i], Cs[i] = dW.data_ptr(), (db.data_ptr() if db is not None else 0)
            else:          # several token chunks: one partial slab per chunk, summed by the batched reducer
                slab = torch.empty(nch, N, K, dtype=torch.float32, device=dev)
                _defer(slab, dW, nch, N * K)
                Cb[i], cstride[i] = slab.data_ptr(), N * K * 4
                cs = None
                if db is not None:
                    cs = torch.empty(nch, N, dtype=torch.float32, device=dev)
                    _defer(cs, db, nch, N)
                    Cs[i], sstride[i] = cs.data_ptr(), N * 4
                keep.append((slab, cs))
        arr = tmpl.copy()
        ch = chunk.astype(np.uint64)
        arr["A"][valid] = A[pidx]
        arr["B"][valid] = B[pidx]
        arr["C"][valid] = Cb[pidx] + ch * cstride[pidx]
        arr["colsum"][valid] = Cs[pidx] + ch * sstride[pidx]
        if WG_STATS is not None:      # dY + X read once, dW written once; 2 tokens N K flop
            WG_STATS.append(("csts_wgrad_grouped8" if rows == 192 else ("csts_wgrad_grouped5" if rows == 96 else "csts_wgrad_grouped"),
                             sum(t[4] * t[5] * t[0].element_size() + t[4] * t[6] * 2 + t[5] * t[6] * 4 for t in probs),
                             sum(2.0 * t[4] * t[5] * t[6] for t in probs)))
        pend.append((arr.tobytes(), n_items, rows, a_f32))
    # the item tables of all tile classes travel in ONE host -> device copy (a copy node per class cost ~12 us each in the replayed
    # step, gap included); one copy per class only when they do not fit the table together
    with (torch.cuda.stream(launch_stream) if launch_stream is not None else contextlib.nullcontext()):
        offs, total = [], 0
        for blob, *_ in pend:
            offs.append(total)
            total += (len(blob) + 255) // 256 * 256
        base = None
        if pend and total <= tab.nbytes:
            base = tab.upload(b"".join(blob + bytes((-len(blob)) % 256) for blob, *_ in pend))
        for (blob, n_items, rows, a_f32), off in zip(pend, offs):
            ptr = base + off if base is not None else tab.upload(blob)
            if rows == 192 and only_w8:
                L.check(_lib().csts_wgrad_grouped8_limited(ptr, n_items, W8_EARLY_WGS, _stream()), "csts_wgrad_grouped8_limited")
            elif rows == 192:
                L.check(_lib().csts_wgrad_grouped8(ptr, n_items, _stream()), "csts_wgrad_grouped8")
            elif rows == 96:
                L.check(_lib().csts_wgrad_grouped5(ptr, n_items, _stream()), "csts_wgrad_grouped5")
            else:
                L.check(_lib().csts_wgrad_grouped(ptr, n_items, 1 if a_f32 else 0, rows, _stream()), "csts_wgrad_grouped")
    if side:
        _wg_pending.append((q, keep))      # alive until the final callback has joined the side stream
    del q, keep


# Stencil weight gradients, grouped (csts_dwconv_wgrad_grouped): 34 first-stage launches of ~24 us per step -- each a single round
# of <= 1024 latency-bound workgroups -- become ONE launch at the end of backward (their second stages were deferred already).
# Same policy as the grouped Linear weight gradients: while the step is captured (GROUP_WGRADS), and only when the gradient may be
# deferred; the operands (the saved q / k / v tensors and the conv-output gradients) stay alive until the flush.
STENCIL_WGRAD_GROUPED = os.environ.get("CSTS_STENCIL_WGRAD_GROUPED", "1") != "0"
# Round 5: the grouped stencil weight gradients (vector-issue-bound, 20 KB of LDS, 124 registers) run on a side stream BESIDE the 128-wide
# classes of the grouped Linear weight gradients (memory-bound, 40 KB of LDS, 154 registers): the two share CUs, -0.2 ... -0.26 ms per step
# (profiles/r5_wgrad_tail_ab.txt).  What made the same side stream neutral in round 4 was the ORDER: the 192 x 384 class went first, and
# its one 144 KB-LDS workgroup per CU leaves no room for anything else -- it now goes last (WGRAD8_LAST).  =0 restores either.
WGRAD8_LAST = os.environ.get("CSTS_WGRAD8_LAST", "1") != "0"
W8_PARALLEL = os.environ.get("CSTS_W8_PARALLEL", "0") == "1"
STENCIL_TAIL_SIDE = os.environ.get("CSTS_STENCIL_TAIL_SIDE", "1") != "0"
_tail_side = {}         # device index -> stream
_swq = []               # (DwconvGeom copy, fine ptr, coarse ptr, workspace ptr, dt, (tensors kept alive))
SWG_EARLY = int(os.environ.get("CSTS_SWG_EARLY", "0"))
_swq_count = [0]        # stencil problems queued so far in this backward pass
_sw_prod = {}           # raw id -> stream: the streams that queued them
_swq_early_keep = []    # operands of an early launch, alive until the final join
_swq_early_used = [False]
_swq_tables = {}        # device index -> HostTable


def _stencil_group_now() -> bool:
    return STENCIL_WGRAD_GROUPED and GROUP_WGRADS != "never" and (GROUP_WGRADS == "always" or torch.cuda.is_current_stream_capturing())


def _queue_stencil_wgrad(g, fine, fine_off, coarse, coarse_off, ws):
    """One csts_dwconv_wgrad problem for the grouped launch of flush_stencil_wgrads (first stage only: the caller defers the row sums
    of `ws` itself)."""
    gg = L.DwconvGeom()
    C.memmove(C.byref(gg), C.byref(g), C.sizeof(L.DwconvGeom))
    _swq.append((gg, _p(fine, fine_off), _p(coarse, coarse_off), _p(ws), _dt(fine), (fine, coarse, ws)))
    cur = torch.cuda.current_stream()
    _sw_prod[cur.cuda_stream] = cur
    _swq_count[0] += 1
    if SWG_EARLY > 0 and _swq_count[0] == SWG_EARLY and fine.is_cuda:
        # Round 5 experiment (CSTS_SWG_EARLY=n): the first n problems of this backward pass (the head's and the late stages' pools: their
        # operands are complete long before the pass ends) go out NOW as one grouped launch on the tail side stream, beside the rest of
        # backward -- ONE fork; the join is the final flush's (the per-pool side launches of round 4 paid 34 fork / join pairs)
        tail = _tail_side.get(cur.device.index)
        if tail is None:
            tail = _tail_side[cur.device.index] = torch.cuda.Stream(device=cur.device)
        for st in _sw_prod.values():
            ev = torch.cuda.Event()
            ev.record(st)
            tail.wait_event(ev)
        with torch.cuda.stream(tail):
            _swq_early_keep.append(flush_stencil_wgrads())
        _swq_early_used[0] = True


def flush_stencil_wgrads():
    if not _swq:
        return None
    q = list(_swq)
    _swq.clear()
    dev = q[0][5][0].device
    tab = _swq_tables.get(dev.index)
    if tab is None:
        tab = _swq_tables[dev.index] = HostTable(L.DWCONV_WGRAD_TABLE_ENTRY * 256, dev, ring=4, captures=16)
    for dt in sorted({e[4] for e in q}):
        sel = [e for e in q if e[4] == dt]
        if len(sel) > 256:
            raise L.CstsError("too many grouped stencil weight gradients")
        items = (L.DwconvWgradItem * len(sel))()
        for i, (gg, fp, cp, wp, _, _) in enumerate(sel):
            items[i].geom, items[i].fine, items[i].coarse, items[i].workspace = gg, fp, cp, wp
        image = (C.c_uint8 * (L.DWCONV_WGRAD_TABLE_ENTRY * len(sel)))()
        nblocks = C.c_int(0)
        L.check(_lib().csts_dwconv_wgrad_grouped_plan(items, len(sel), image, len(image), C.byref(nblocks)), "csts_dwconv_wgrad_grouped_plan")
        ptr = tab.upload(bytes(image))
        if WG_STATS is not None:          # fine + coarse read once; 2 x 27 flop per coarse cell and channel
            esz = 4 if dt == F32 else 2
            cells = lambda gg: (gg.B * gg.Tf * gg.Hf * gg.Wf, gg.B * gg.Tc * gg.Hc * gg.Wc)
            WG_STATS.append(("csts_dwconv_wgrad_grouped", sum(sum(cells(e[0])) * e[0].C * esz for e in sel),
                             sum(54.0 * cells(e[0])[1] * e[0].C for e in sel)))
        L.check(_lib().csts_dwconv_wgrad_grouped(ptr, len(sel), nblocks.value, dt, _stream()), "csts_dwconv_wgrad_grouped")
    return q            # the operands: the caller keeps them alive until the launch stream has been joined


# Stencil weight gradients (csts_dwconv_wgrad / _wgrad2: 34 launches of ~24 us per step, latency-bound, 0.16 of the HBM roofline)
# feed nothing but the end-of-backward reduction: with deferred reductions they run on a SIDE stream beside the rest of backward
# (one side stream per producing stream: the audio trunk's backward has its own) and are joined by the final flush.
STENCIL_WGRAD_SIDE = os.environ.get("CSTS_STENCIL_WGRAD_SIDE", "0") == "1"     # MEASURED SLOWER (21.3 -> 22.2 ms per step, profiles/r4_stencil_wgrad_side_ab.txt): 34 fork / join pairs inside the captured graph cost more than the overlap returns; off
_sw_side = {}            # raw id of the producing stream -> side stream
_sw_used = [False]
_sw_keep = []            # tensors read / written on a side stream, alive until the join


def _stencil_side(*tensors):
    """Side stream for a stencil weight-gradient launch issued from the current stream, already waiting for everything queued on
    it so far; `tensors` (operands, workspace) are kept alive and marked as used there."""
    cur = torch.cuda.current_stream()
    st = _sw_side.get(cur.cuda_stream)
    if st is None:
        st = _sw_side[cur.cuda_stream] = torch.cuda.Stream(device=cur.device)
    st.wait_stream(cur)
    for t in tensors:
        if t is not None:
            t.record_stream(st)
            _sw_keep.append(t)
    _sw_used[0] = True
    return st


def _join_stencil_side():
    if _sw_used[0]:
        cur = torch.cuda.current_stream()
        for st in _sw_side.values():
            cur.wait_stream(st)
        _sw_used[0] = False
    _sw_keep.clear()


def _join_wgrad_side():
    """Make the current stream wait for the side-stream weight-gradient launches of this backward pass."""
    if _wg_side_used[0]:
        cur = torch.cuda.current_stream()
        for st in _wg_side.values():
            cur.wait_stream(st)
        _wg_side_used[0] = False
    _wg_pending.clear()


# ----------------------------------------------------------------------------------------- raw wrappers
AUTO_SPLIT_SMALL_M = True
MID_M_SPLIT = int(os.environ.get("CSTS_MID_M_SPLIT", "1"))


def _auto_split(layout, M, N, K, compute):
    """Deterministic split-K for activation GEMMs with a handful of rows (embedding projections and similarity matrices
    of the EgoNCE loss: M = batch; the temporal-fusion block: M = B*2T'): one or two row tiles leave 1..24 workgroups
    walking the whole K serially (55-100 us per call measured); slabs + the finishing pass spread K over the chip."""
    if layout != L.GEMM_TN and compute == BF16 and MID_M_SPLIT > 1 and 1024 <= M <= 4096 and K >= 2304 and N <= 1024:
        # the 768-channel stage at b = 4 (M = 2048 tokens): fc2 / the qkv data gradient are 64-96 output tiles on 256 CUs with a
        # 36-48 step k-loop each -- cut the loop so that every CU has a tile (deterministic slabs + finishing pass)
        return MID_M_SPLIT
    if layout == L.GEMM_TN or M > 128:
        return 1
    ksteps = -(-K // (64 if compute == BF16 else 32))
    tiles = -(-M // 64) * -(-N // 128)
    split = min(ksteps // (4 if compute == BF16 else 2), -(-192 // tiles))
    return split if split >= 2 else 1


def gemm(layout, A, a_off, lda, B, b_off, ldb, Cm, ldc, M, N, K, *, compute, bias=None, epilogue=L.EPI_NONE, aux=None,
         residual=None, ldr=0, res_row_mod=0, row_scale=None, rows_per_scale=1, split_k=1, deterministic=True, tile_rows=0,
         want_colsum=False, algo=0, debug_ws=None, res_up=None):
    """res_up = (coarse thw, fine thw): `residual` is the coarse token grid and the epilogue adds its trilinear up-sampling
    (csts_gemm_args.res_up)."""
    if split_k == 1 and AUTO_SPLIT_SMALL_M and not want_colsum and algo == 0 and tile_rows == 0 and res_up is None:
        split_k = _auto_split(layout, M, N, K, compute)
    a = L.GemmArgs()
    a.layout = layout
    a.A, a.a_dt, a.lda = _p(A, a_off), _dt(A), lda
    a.B, a.b_dt, a.ldb = _p(B, b_off), _dt(B), ldb
    a.C, a.c_dt, a.ldc = _p(Cm), _dt(Cm), ldc
    a.M, a.N, a.K = M, N, K
    a.bias = _p(bias)
    a.epilogue = epilogue
    # GELU without a kept pre-activation (inference): aux_dt still names the activation dtype -- it selects the GELU flavour
    # (exact erf in fp32 mode, the 1.5e-7 polynomial in the 16-bit modes), which must not depend on whether h is kept
    a.aux, a.aux_dt, a.ldaux = _p(aux), (_dt(aux) if aux is not None else (_dt(Cm) if epilogue == L.EPI_GELU else 0)), (N if aux is not None else 0)
    a.residual, a.r_dt, a.ldr = _p(residual), (_dt(residual) if residual is not None else 0), ldr
    a.res_row_mod = res_row_mod
    a.row_scale, a.rows_per_scale = _p(row_scale), rows_per_scale
    a.compute, a.split_k = compute, split_k
    a.tile_rows = tile_rows
    a.algo = algo
    if res_up is not None:
        for i, v in enumerate(list(res_up[0]) + list(res_up[1])):
            a.res_up[i] = int(v)
    ws = None
    if debug_ws is not None:   # diagnostics builds only (gemm3 cycle stamps)
        a.workspace, a.ws_bytes = _p(debug_ws), debug_ws.numel() * debug_ws.element_size()
    if split_k > 1 and deterministic:
        nb = _lib().csts_gemm_splitk_workspace(M, N, K, split_k)
        if nb <= SPLITK_WS_LIMIT:
            ws = _ws(nb, Cm.device)
            a.workspace, a.ws_bytes = _p(ws), ws.numel()
    fused = None
    if want_colsum and (split_k == 1 or ws is not None) and _lib().csts_gemm_v2_eligible(C.byref(a)):
        fused = torch.empty(M, dtype=torch.float32, device=Cm.device)
        a.colsum = _p(fused)
    L.check(_lib().csts_gemm(C.byref(a), _stream()), "csts_gemm")
    return fused


def colsum(X: torch.Tensor, batch: int, M: int, N: int, row_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty(batch * N, dtype=torch.float32, device=X.device)
    nbytes = _lib().csts_colsum_workspace(batch, M, N)
    ws = _ws(nbytes, X.device)
    L.check(_lib().csts_colsum(_p(X), _dt(X), _p(row_weight), _p(out), batch, M, N, _p(ws), ws.numel(), _stream()),
            "csts_colsum")
    return out


def scale_rows(x: torch.Tensor, row_scale: torch.Tensor, rows_per_scale: int, M: int, N: int, out_dt=None) -> torch.Tensor:
    out = torch.empty_like(x) if out_dt is None else torch.empty(x.shape, dtype=torch_dtype(out_dt), device=x.device)
    L.check(_lib().csts_scale_rows(_p(x), _dt(x), _p(row_scale), rows_per_scale, _p(out), _dt(out), M, N, _stream()),
            "csts_scale_rows")
    return out


SPLITK_WS_LIMIT = 256 << 20   # deterministic split-K slabs up to this size, atomics beyond


def _wgrad_split(M_out: int, N_out: int, Kred: int) -> int:
    tiles = math.ceil(M_out / 128) * math.ceil(N_out / 128)
    return max(1, min(512 // max(tiles, 1), math.ceil(Kred / 512)))


def _wgrad(dY: torch.Tensor, X: torch.Tensor, M: int, N: int, K: int, compute: int, want_bias: bool = False,
           params=None):
    """dW[N,K] = dY[M,N]^T X[M,K] (fp32, split over M); with want_bias also db[N] = colsum(dY), fused into the same
    kernel when the bf16 v2 GEMM applies.  params = (W, b) leaf parameters: inside a backward pass the problem may then be
    queued for the grouped end-of-backward launch, which assigns the gradients itself -- the return value is None (or
    (None, None)) in that case and the caller returns None to autograd for those inputs."""
    if compute == BF16 and params is not None:
        if queue_wgrad(dY, X, M, N, K, params[0], params[1] if want_bias else None):
            return (None, None) if want_bias else None
    split = _wgrad_split(N, K, M)
    det = split > 1 and _lib().csts_gemm_splitk_workspace(N, K, M, split) <= SPLITK_WS_LIMIT
    dW = (torch.zeros if (split > 1 and not det) else torch.empty)(N, K, dtype=torch.float32, device=dY.device)
    db = gemm(L.GEMM_TN, dY, 0, N, X, 0, K, dW, K, N, K, M, compute=compute, split_k=split, deterministic=det,
              want_colsum=want_bias)
    if want_bias:
        return dW, (db if db is not None else colsum(dY, 1, M, N))
    return dW


# ----------------------------------------------------------------------------------------- LayerNorm
BF16_GRAD_COPY = os.environ.get("CSTS_BF16_GRAD_COPY", "1") != "0"


def _ln_bwd_call(dy, x, gamma, mean, rstd, addend, rows, Cc, what, want16=False, params=None, copy_scale=None, dy2=None):
    """dx (+ addend) and the [2*C] dgamma|dbeta buffer.  params = (gamma, beta) leaf parameters: inside a backward pass
    the second stage is then deferred and the flush assigns the two halves itself -- the returned buffer is None.
    want16: also emit a bf16 copy of dx, attached as dx._csts_bf16, for the weight/data-gradient GEMMs that read it next
    (LinearFn / MlpFn backward pick it up; any other consumer simply ignores the attribute)."""
    dx = torch.empty_like(x)
    dx16 = torch.empty(x.shape, dtype=L.half_dtype(), device=x.device) if (want16 and BF16_GRAD_COPY) else None
    dgb = torch.empty(2 * Cc, dtype=torch.float32, device=x.device)
    nbytes = _lib().csts_layernorm_bwd_workspace(rows, Cc)
    ws = _ws(nbytes, x.device)
    defer = params is not None and _can_defer(*params)
    if dx16 is None or copy_scale is None or rows % copy_scale[1] != 0 or copy_scale[0].numel() * copy_scale[1] != rows:
        copy_scale = None
    cs, rps = (copy_scale[0], copy_scale[1]) if copy_scale is not None else (None, 1)
    L.check(_lib().csts_layernorm_bwd_ex(_p(dy), _p(dy2), _dt(dy), _p(x), _dt(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _dt(dx),
                                         _p(addend), _p(dx16), _p(cs), rps, None if defer else _p(dgb),
                                         None if defer else _p(dgb, Cc), _p(ws), ws.numel(), rows, Cc, _stream()), what)
    if defer:
        _defer(ws, dgb, nbytes // (2 * Cc * 4), 2 * Cc)
        _assign_later(params[0], dgb[:Cc])
        _assign_later(params[1], dgb[Cc:])
        dgb = None
    if dx16 is not None:
        _attach16(dx, dx16, copy_scale)
    return dx, dgb


def _scale_key(scale):
    """Identity of a (row_scale tensor, rows_per_scale) pair: saved tensors come back as new Python objects, so the storage
    address stands for the tensor."""
    if scale is None:
        return None
    return (scale[0].data_ptr(), scale[0].numel(), int(scale[1]))


def _attach16(dx: torch.Tensor, dx16: torch.Tensor, scale=None):
    """Remember the bf16 copy of the fp32 gradient `dx` on the tensor object, together with dx's version counter: autograd
    accumulates the gradients of a tensor with several consumers IN PLACE into the first one that arrives, which keeps
    the Python object (and this attribute) but bumps the version -- the copy is then stale and must not be used.
    scale = (row_scale, rows_per_scale): the copy holds dx times that per-sample scale (csts_layernorm_bwd_ex)."""
    dx._csts_bf16 = (dx16, dx._version, _scale_key(scale))


def _grad16(dy: torch.Tensor, compute: int, scale=None):
    """The bf16 copy of a residual-stream gradient left by the kernel that produced it (or None: no copy, `dy` was
    modified since the copy was made, or the copy was not made with the per-sample scale the caller needs)."""
    if compute != BF16 or dy.dtype != torch.float32:
        return None
    tag = getattr(dy, "_csts_bf16", None)
    if tag is None:
        return None
    d16, ver, key = tag
    if dy._version != ver or d16.shape != dy.shape or d16.device != dy.device or key != _scale_key(scale):
        return None
    return d16


def _mark_scaled_output(y, row_scale, rows_per_scale):
    """y = f(...) * row_scale + residual: remember the scale on the output tensor, so that the LayerNorm that reads y next can
    leave the bf16 copy of dL/dy already multiplied by it (LayerNormFn / _ln_bwd_call)."""
    if row_scale is not None and SCALED_GRAD_COPY:
        y._csts_prod_scale = (row_scale, int(rows_per_scale))
    return y


SCALED_GRAD_COPY = os.environ.get("CSTS_SCALED_GRAD_COPY", "1") != "0"


class LayerNormFn(Function):
    """nn.LayerNorm over the last dim (attention.py:192,214 eps 1e-6; :108,112,116 eps 1e-5).

    With ``passthrough`` the input is returned as a second output (an alias): the pre-norm residual pattern
    x + f(LN(x)) (attention.py:242,247) then hands BOTH gradients of x to this node, and the backward kernel writes
    dx = LN'(d_xn) + d_residual in one pass instead of leaving the sum to a separate autograd add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, out_dt: int, passthrough: bool, fanout: bool = False, addend=None):
        _need_gpu(x, gamma, beta)
        if passthrough and not x.is_contiguous():
            raise L.CstsError("layer_norm(passthrough=True) needs a contiguous input")
        x_in = x
        x = x.contiguous()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        y = torch.empty(x.shape, dtype=torch_dtype(out_dt), device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ctx.has_add = addend is not None
        if addend is not None:              # LN(x + addend); the sum (the new residual stream) is the pass-through output
            if not passthrough or addend.shape != x.shape or addend.dtype != x.dtype:
                raise L.CstsError("layer_norm(addend=...) needs passthrough=True and an addend like x")
            addend = addend.contiguous()
            xs = torch.empty_like(x)
            L.check(_lib().csts_layernorm_fwd_add(_p(x), _p(addend), _p(xs), _dt(x), _p(gamma), _p(beta), _p(y), out_dt, _p(mean),
                                                  _p(rstd), rows, Cc, eps, _stream()), "csts_layernorm_fwd_add")
            x = xs
        else:
            L.check(_lib().csts_layernorm_fwd(_p(x), _dt(x), _p(gamma), _p(beta), _p(y), out_dt, _p(mean), _p(rstd), rows, Cc,
                                              eps, _stream()), "csts_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.copy_scale = getattr(x_in, "_csts_prod_scale", None) if passthrough else None
        ctx.set_materialize_grads(False)
        ctx.fanout = bool(fanout)
        if fanout:                          # two aliases of y, one per consumer: backward receives both gradients
            if not passthrough:
                raise L.CstsError("layer_norm(fanout=True) comes with passthrough=True")
            return y, y.view_as(y), x
        if passthrough:
            return y, x
        return y

    @staticmethod
    def backward(ctx, dy, *rest):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy2 = None
        if ctx.fanout:
            dy2, dpass = rest
            if dy is None:
                dy, dy2 = dy2, None
        else:
            dpass = rest[0] if rest else None
        nret = 8
        # LN(x + addend): x and the addend receive the same gradient (d(x + a)/dx = d(x + a)/da = 1)
        both = (lambda g_: (g_,) + (None,) * (nret - 2) + (g_,)) if ctx.has_add else (lambda g_: (g_,) + (None,) * (nret - 1))
        if dy is None:                      # only the residual branch carried a gradient
            return both(dpass)
        dy = dy.contiguous()
        if dy2 is not None:
            dy2 = dy2.contiguous()
            if dy2.dtype != dy.dtype or dy2.shape != dy.shape:
                dy, dy2 = dy + dy2, None
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        if dpass is not None:
            dpass = dpass.contiguous()
            if dpass.dtype != x.dtype:
                dpass = dpass.to(x.dtype)
        dx, dgb = _ln_bwd_call(dy, x, gamma, mean, rstd, dpass, rows, Cc, "csts_layernorm_bwd",
                               want16=(x.dtype == torch.float32 and dy.dtype == L.half_dtype()), params=ctx.params,
                               copy_scale=ctx.copy_scale, dy2=dy2)
        if dgb is None:                     # finished and assigned by the end-of-backward flush
            return both(dx)
        return (dx, dgb[:Cc], dgb[Cc:]) + (None,) * (nret - 4) + ((dx,) if ctx.has_add else (None,))


def layer_norm(x, gamma, beta, eps, out_dt, passthrough: bool = False, fanout: bool = False, addend=None):
    """passthrough=False: y.  passthrough=True: (y, x_alias) -- use x_alias wherever x is used afterwards.
    fanout=True (with passthrough): (y_a, y_b, x_alias) for a LayerNorm output with TWO consumers -- the backward kernel reads
    both incoming gradients itself (csts_layernorm_bwd_ex) instead of autograd adding them first.
    addend (with passthrough): LayerNorm(x + addend), and x_alias is that sum (csts_layernorm_fwd_add) -- a skip connection
    folded into the norm that follows it."""
    return LayerNormFn.apply(x, gamma, beta, eps, out_dt, passthrough, fanout, addend)


# ----------------------------------------------------------------------------------------- Linear
class LinearFn(Function):
    """y = (x W^T + b) * row_scale + residual  (nn.Linear: attention.py:130,159,232,246; drop_path + residual
    of attention.py:242,247 folded into the epilogue)."""

    @staticmethod
    def forward(ctx, x, W, b, residual, row_scale, rows_per_scale: int, out_dt: int, compute: int, w16, w16t=None, res_up=None):
        """res_up = (coarse thw, fine thw): `residual` is the decoder skip on the COARSE grid and the GEMM epilogue adds
        nn.Upsample(scale_factor=stride_q, mode='trilinear')(residual) (attention.py:463-471) -- x_res is never written."""
        _need_gpu(x, W)
        x = x.contiguous()
        W = W.contiguous()
        K = x.shape[-1]
        N = W.shape[0]
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], N, dtype=torch_dtype(out_dt), device=x.device)
        if residual is not None:
            residual = residual.contiguous()
        Wop = w16 if (w16 is not None and compute == BF16) else W     # bf16 shadow of the fp32 master weight
        gemm(L.GEMM_NT, x, 0, K, Wop, 0, K, y, N, M, N, K, compute=compute, bias=b, residual=residual, ldr=N,
             row_scale=row_scale, rows_per_scale=rows_per_scale, res_up=res_up)
        ctx.res_up = None
        if res_up is not None:
            thw_c, thw_f = list(res_up[0]), list(res_up[1])
            Bc = residual.shape[0]
            ctx.res_up = (_pool_geom(Bc, N, thw_c, thw_f, [f // c for f, c in zip(thw_f, thw_c)]), tuple(residual.shape))
        ctx.save_for_backward(x, Wop, row_scale)
        ctx.wdtype = W.dtype
        ctx.params = (W, b)
        ctx.w16t = w16t if (w16t is not None and compute == BF16 and Wop is w16) else None
        ctx.meta = (M, N, K, rows_per_scale, compute, b is not None, residual is not None,
                    residual.dtype if residual is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, row_scale = ctx.saved_tensors
        M, N, K, rps, compute, has_b, has_res, res_dtype = ctx.meta
        sc = (row_scale, rps) if row_scale is not None else None
        d16 = _grad16(dy, compute, sc) if dy.is_contiguous() else None     # with drop-path: the copy pre-multiplied by it
        dy = dy.contiguous()
        if d16 is not None:
            dys = d16
        elif row_scale is not None:        # drop-path: the scaled copy goes straight to bf16 when the GEMMs run in bf16
            dys = scale_rows(dy, row_scale, rps, M, N, out_dt=BF16 if (compute == BF16 and BF16_GRAD_COPY) else None)
        else:
            dys = dy
        dx = dW = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _dgrad(dys, W, ctx.w16t, dx, M, N, K, compute)
        if ctx.needs_input_grad[1]:
            if has_b and ctx.needs_input_grad[2]:
                dW, db = _wgrad(dys, x, M, N, K, compute, want_bias=True, params=ctx.params)
            else:
                dW = _wgrad(dys, x, M, N, K, compute, params=(ctx.params[0], None))
            if dW is not None:
                dW = dW.to(ctx.wdtype)
        elif has_b and ctx.needs_input_grad[2]:
            db = colsum(dys, 1, M, N)
        if has_res and ctx.needs_input_grad[3]:
            if ctx.res_up is not None:      # adjoint of the up-sampling the epilogue applied to the coarse skip
                g, shape = ctx.res_up
                dres = torch.empty(shape, dtype=res_dtype, device=dy.device)
                L.check(_lib().csts_trilinear_bwd(C.byref(g), _p(dy), _dt(dy), _p(dres), _dt(dres), _stream()), "csts_trilinear_bwd(res_up)")
            else:
                dres = dy if dy.dtype == res_dtype else dy.to(res_dtype)
        return dx, dW, db, dres, None, None, None, None, None, None, None


def _dgrad(dy, W, Wt, dx, M, N, K, compute, epilogue=L.EPI_NONE, aux=None):
    """dx[M,K] = dy[M,N] W[N,K].  With the [K][N] twin of the bf16 shadow (csts_transpose_multi) and a bf16 dy this is an
    NT GEMM -- both operands contiguous along the reduction, the forward's kernels -- else NN on W itself."""
    if Wt is not None and dy.dtype == L.half_dtype() and USE_W16T:
        gemm(L.GEMM_NT, dy, 0, N, Wt, 0, N, dx, K, M, K, N, compute=compute, epilogue=epilogue, aux=aux)
    else:
        gemm(L.GEMM_NN, dy, 0, N, W, 0, K, dx, K, M, K, N, compute=compute, epilogue=epilogue, aux=aux)


USE_W16T = os.environ.get("CSTS_W16T", "1") != "0"


class _TransposeSet:
    """The [in][out] twins of a fixed set of bf16 matrices, refreshed by ONE csts_transpose_multi launch (tile table built
    once: the buffers never move)."""

    def __init__(self, pairs):
        self.pairs = list(pairs)                 # (src [R, C] bf16, dst [C, R] bf16)
        tiles = []
        for src, dst in self.pairs:
            R, Cc = src.shape
            assert src.dtype == L.half_dtype() and dst.dtype == L.half_dtype() and tuple(dst.shape) == (Cc, R)
            assert R % 8 == 0 and Cc % 8 == 0 and src.is_contiguous() and dst.is_contiguous()
            for r0 in range(0, R, 64):
                for c0 in range(0, Cc, 64):
                    tiles.append((src.data_ptr(), dst.data_ptr(), R, Cc, r0, c0))
        arr = (L.TransposeTile * len(tiles))()
        for i, t in enumerate(tiles):
            arr[i].src, arr[i].dst, arr[i].R, arr[i].C, arr[i].r0, arr[i].c0 = t
        self.n = len(tiles)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.pairs[0][0].device)

    def refresh(self):
        L.check(_lib().csts_transpose_multi(self.table.data_ptr(), self.n, _stream()), "csts_transpose_multi")


RES_UP_FUSE = os.environ.get("CSTS_RES_UP_FUSE", "1") != "0"     # decoder skip up-sampled inside the proj GEMM's epilogue


def res_up_ok(thw_fine) -> bool:
    """The fused form needs power-of-two fine grid sizes (csts_gemm_args.res_up)."""
    return RES_UP_FUSE and all(v > 0 and (v & (v - 1)) == 0 for v in thw_fine)


def linear(x, W, b=None, *, residual=None, row_scale=None, rows_per_scale=1, out_dt=F32, compute=F32, w16=None, w16t=None,
           res_up=None):
    y = LinearFn.apply(x, W, b, residual, row_scale, rows_per_scale, out_dt, compute, w16, w16t, res_up)
    return _mark_scaled_output(y, row_scale if residual is not None else None, rows_per_scale)


class MlpFn(Function):
    """fc2(GELU_erf(fc1(x))) * row_scale + residual  (Mlp.forward common.py:26-34; attention.py:244-247).
    GELU lives in fc1's epilogue, GELU' in the epilogue of fc2's data-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, residual, row_scale, rows_per_scale: int, act_dt: int, out_dt: int, compute: int,
                w16_1, w16_2, w16t_1=None, w16t_2=None):
        _need_gpu(x, W1, W2)
        ctx.params = (W1, b1, W2, b2)
        ctx.w16t = (None, None)
        if compute == BF16 and w16_1 is not None and w16_2 is not None:   # bf16 shadows of the fp32 master weights
            W1, W2 = w16_1, w16_2
            ctx.w16t = (w16t_1, w16t_2)
        x = x.contiguous()
        K = x.shape[-1]
        Hd = W1.shape[0]
        N = W2.shape[0]
        M = x.numel() // K
        # the pre-activation h is kept for backward only: an inference forward (no input needs a gradient) does not write it
        # (M x 4C x 2 bytes per block: 1.26 GB of a b = 4 forward at 16 x 256^2)
        train = any(ctx.needs_input_grad)
        g = torch.empty(M, Hd, dtype=torch_dtype(act_dt), device=x.device)
        h = torch.empty_like(g) if train else None   # pre-activation
        gemm(L.GEMM_NT, x, 0, K, W1, 0, K, g, Hd, M, Hd, K, compute=compute, bias=b1, epilogue=L.EPI_GELU, aux=h)
        y = torch.empty(*x.shape[:-1], N, dtype=torch_dtype(out_dt), device=x.device)
        if residual is not None:
            residual = residual.contiguous()
        gemm(L.GEMM_NT, g, 0, Hd, W2, 0, Hd, y, N, M, N, Hd, compute=compute, bias=b2, residual=residual, ldr=N,
             row_scale=row_scale, rows_per_scale=rows_per_scale)
        if train:
            ctx.save_for_backward(x, W1, W2, h, g, row_scale)
        ctx.meta = (M, K, Hd, N, rows_per_scale, compute, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W1, W2, h, g, row_scale = ctx.saved_tensors
        M, K, Hd, N, rps, compute, has_res = ctx.meta
        sc = (row_scale, rps) if row_scale is not None else None
        d16 = _grad16(dy, compute, sc) if dy.is_contiguous() else None     # with drop-path: the copy pre-multiplied by it
        dy = dy.contiguous()
        if d16 is not None:
            dys = d16
        elif row_scale is not None:
            dys = scale_rows(dy, row_scale, rps, M, N, out_dt=BF16 if (compute == BF16 and BF16_GRAD_COPY) else None)
        else:
            dys = dy
        P1, pb1, P2, pb2 = ctx.params
        dW2, db2 = _wgrad(dys, g, M, N, Hd, compute, want_bias=True, params=(P2, pb2))
        dh = torch.empty_like(h)
        _dgrad(dys, W2, ctx.w16t[1], dh, M, N, Hd, compute, epilogue=L.EPI_DGELU, aux=h)
        dW1, db1 = _wgrad(dh, x, M, Hd, K, compute, want_bias=True, params=(P1, pb1))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _dgrad(dh, W1, ctx.w16t[0], dx, M, Hd, K, compute)
        return dx, dW1, db1, dW2, db2, (dy if has_res else None), None, None, None, None, None, None, None, None, None


def mlp(x, W1, b1, W2, b2, *, residual=None, row_scale=None, rows_per_scale=1, act_dt=F32, out_dt=F32, compute=F32,
        w16_1=None, w16_2=None, w16t_1=None, w16t_2=None):
    y = MlpFn.apply(x, W1, b1, W2, b2, residual, row_scale, rows_per_scale, act_dt, out_dt, compute, w16_1, w16_2,
                    w16t_1, w16t_2)
    return _mark_scaled_output(y, row_scale if residual is not None else None, rows_per_scale)


# ----------------------------------------------------------------------------------------- attention inner
def _conv_geom(B, Cc, HD, fine_thw, stride, f_bs, f_ts, c_bs, c_ts) -> L.DwconvGeom:
    g = L.DwconvGeom()
    g.B, g.C, g.HD = B, Cc, HD
    g.Tf, g.Hf, g.Wf = fine_thw
    g.st, g.sh, g.sw = stride
    g.Tc, g.Hc, g.Wc = [(f - 1) // s + 1 for f, s in zip(fine_thw, stride)]
    g.fine_batch_stride, g.fine_token_stride = f_bs, f_ts
    g.coarse_batch_stride, g.coarse_token_stride = c_bs, c_ts
    return g


def _ln_rows_fwd(c, gamma, beta, HD, act_dt):
    rows = c.numel() // HD
    y = torch.empty(c.shape, dtype=torch_dtype(act_dt), device=c.device)
    mean = torch.empty(rows, dtype=torch.float32, device=c.device)
    rstd = torch.empty_like(mean)
    L.check(_lib().csts_layernorm_fwd(_p(c), _dt(c), _p(gamma), _p(beta), _p(y), act_dt, _p(mean), _p(rstd), rows, HD, 1e-5,
                                      _stream()), "csts_layernorm_fwd(head)")
    return y, mean, rstd


def _ln_rows_bwd(dy, c, gamma, mean, rstd, HD, params=None):
    rows = c.numel() // HD
    dc, dgb = _ln_bwd_call(dy, c, gamma, mean, rstd, None, rows, HD, "csts_layernorm_bwd(head)", params=params)
    if dgb is None:
        return dc, None, None
    return dc, dgb[:HD], dgb[HD:]


class AttnInnerFn(Function):
    """Everything between the qkv Linear and the output projection of MultiScaleAttention /
    MultiScaleDecoderAttention / SpatialAttention / TemporalAttention (attention.py:130-158, :374-388;
    av_attention.py:127-141, :324-350): conv-pool (or ConvTranspose upsample of q) + LayerNorm(hd) of
    q/k/v read straight from the token-major qkv buffer, then the fused attention core.
    kind: 'enc' | 'dec' | 'plain'.  Returns o (B, Nq, C)."""

    @staticmethod
    def forward(ctx, qkv, kvc, wq, gq, bq, wk, gk, bk, wv, gv, bv, meta):
        """kvc is None: qkv is the (B, N, 3C) output of the qkv Linear.  kvc given (qkv_compact): `qkv` holds q only, (B, N, C),
        and kvc = k|v (B, Nc, 2C) on the COMPACT grid of the rows the K/V pools read (csts_kv_rows_geom): the pools then run
        over that grid with stride (1, 3, 3) -- the same taps, the same arithmetic."""
        _need_gpu(qkv)
        (B, N, Cc, H, thw, kind, stride_q, stride_kv, has_pool_q, has_pool_kv, mask_mode, mask_T, mask_HW, act_dt) = meta
        qkv = qkv.contiguous()
        HD = Cc // H
        dev = qkv.device
        lib = _lib()
        s = _stream()
        saved = {}
        QW = Cc if kvc is not None else 3 * Cc          # row width of the tensor that holds q
        if kvc is not None:
            kvc = kvc.contiguous()
            kv_thw = [thw[0]] + list(kv_compact_dims(thw, stride_kv))
            assert has_pool_kv and kv_thw[0] * kv_thw[1] * kv_thw[2] == kvc.shape[1] and kvc.shape[2] == 2 * Cc
            kv_src, kv_off, kv_w, kv_n, kv_stride = kvc, (0, Cc), 2 * Cc, kvc.shape[1], (1, 3, 3)
        else:
            kv_thw, kv_src, kv_off, kv_w, kv_n, kv_stride = list(thw), qkv, (Cc, 2 * Cc), 3 * Cc, N, stride_kv

        def slot_view(slot):   # (tensor, elem offset, strides) of q/k/v inside the qkv buffer
            return (qkv, slot * Cc, (N * QW, QW, HD), N)

        def pooled(slot, w, gamma, beta, stride, transposed):
            if transposed:   # decoder q: ConvTranspose3d, fine = output grid
                fine_thw = [t * st for t, st in zip(thw, stride)]
                Nf = fine_thw[0] * fine_thw[1] * fine_thw[2]
                g = _conv_geom(B, Cc, HD, fine_thw, stride, Nf * Cc, Cc, N * QW, QW)
                c = torch.empty(B, Nf, Cc, dtype=qkv.dtype, device=dev)
                L.check(lib.csts_dwconv_transposed(C.byref(g), _p(qkv, slot * Cc), _dt(qkv), _p(w), _p(c), _dt(c), s),
                        "csts_dwconv_transposed")
                n_out = Nf
                out_thw = fine_thw
            else:
                g = _conv_geom(B, Cc, HD, list(thw), stride, N * QW, QW, 0, Cc)
                n_out = g.Tc * g.Hc * g.Wc
                g.coarse_batch_stride = n_out * Cc
                c = torch.empty(B, n_out, Cc, dtype=qkv.dtype, device=dev)
                L.check(lib.csts_dwconv_strided(C.byref(g), _p(qkv, slot * Cc), _dt(qkv), _p(w), _p(c), _dt(c), s),
                        "csts_dwconv_strided")
                out_thw = [g.Tc, g.Hc, g.Wc]
            y, mean, rstd = _ln_rows_fwd(c, gamma, beta, HD, act_dt)
            saved[slot] = (c, mean, rstd, g)
            return (y, 0, (n_out * Cc, Cc, HD), n_out), out_thw

        def pooled_fused(slots, ws_, gammas, betas, stride):
            """Conv-pool + LayerNorm(hd) of 1 or 2 slots that share the geometry in ONE launch (csts_pool_ln_fwd)."""
            ns = len(slots)
            if slots[0] == 0:       # q: from the tensor that holds q
                src, offs, g = qkv, (0,), _conv_geom(B, Cc, HD, list(thw), stride, N * QW, QW, 0, Cc)
            else:                   # k | v: from the qkv buffer, or from the compact k|v tensor with the compact grid's stride
                src, offs, g = kv_src, kv_off, _conv_geom(B, Cc, HD, kv_thw, kv_stride, kv_n * kv_w, kv_w, 0, Cc)
            n_out = g.Tc * g.Hc * g.Wc
            g.coarse_batch_stride = n_out * Cc
            rows = B * n_out * H
            c = torch.empty(ns, B, n_out, Cc, dtype=qkv.dtype, device=dev)
            y = torch.empty_like(c)
            mean = torch.empty(ns, rows, dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            pa = L.PoolLnArgs()
            pa.geom, pa.nslots, pa.dt, pa.eps = g, ns, _dt(qkv), 1e-5
            for i, slot in enumerate(slots):
                pa.fine[i], pa.weight[i], pa.gamma[i], pa.beta[i] = _p(src, offs[i]), _p(ws_[i]), _p(gammas[i]), _p(betas[i])
                pa.conv_out[i], pa.y[i], pa.mean[i], pa.rstd[i] = _p(c[i]), _p(y[i]), _p(mean[i]), _p(rstd[i])
            L.check(lib.csts_pool_ln_fwd(C.byref(pa), s), "csts_pool_ln_fwd")
            for i, slot in enumerate(slots):
                saved[slot] = (c[i], mean[i], rstd[i], g)
            if ns == 2:
                saved["kv"] = (c, mean, rstd, g)       # stacked pair: one-launch backward kernels
            return [(y[i], 0, (n_out * Cc, Cc, HD), n_out) for i in range(ns)]

        if kind == "dec":
            qd, _ = pooled(0, wq, gq, bq, stride_q, True)
        elif has_pool_q:
            qd, = pooled_fused([0], [wq], [gq], [bq], stride_q)
        else:
            qd = slot_view(0)
        if has_pool_kv:
            kd, vd = pooled_fused([1, 2], [wk, wv], [gk, gv], [bk, bv], stride_kv)
        else:
            kd, vd = slot_view(1), slot_view(2)
        Nq, Nk = qd[3], kd[3]
        o = torch.empty(B, Nq, Cc, dtype=qkv.dtype, device=dev)
        lse = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
        a = L.AttnArgs()
        a.Q, a.K, a.V = _p(qd[0], qd[1]), _p(kd[0], kd[1]), _p(vd[0], vd[1])
        a.O, a.LSE = _p(o), _p(lse)
        a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = _dt(qkv), B, H, Nq, Nk, HD
        a.q_strides = (C.c_int64 * 3)(*qd[2])
        a.k_strides = (C.c_int64 * 3)(*kd[2])
        a.v_strides = (C.c_int64 * 3)(*vd[2])
        a.o_strides = (C.c_int64 * 3)(Nq * Cc, Cc, HD)
        a.scale = HD ** -0.5
        a.mask_mode, a.mask_T, a.mask_HW = mask_mode, mask_T, mask_HW
        L.check(lib.csts_attn_fwd(C.byref(a), s), "csts_attn_fwd")
        ctx.meta = meta
        ctx.saved_slots = saved
        ctx.descr = (qd, kd, vd, Nq, Nk)
        ctx.save_for_backward(qkv, o, lse, wq, gq, wk, gk, wv, gv, kvc)
        ctx.params = (wq, gq, bq, wk, gk, bk, wv, gv, bv)
        ctx.mark_non_differentiable(lse)
        ctx.set_materialize_grads(False)     # no zero-filled d(lse) tensor per backward
        return o, lse

    @staticmethod
    def backward(ctx, do, _dlse):
        (B, N, Cc, H, thw, kind, stride_q, stride_kv, has_pool_q, has_pool_kv, mask_mode, mask_T, mask_HW, act_dt) = ctx.meta
        qkv, o, lse, wq, gq, wk, gk, wv, gv, kvc = ctx.saved_tensors
        qd, kd, vd, Nq, Nk = ctx.descr
        saved = ctx.saved_slots
        HD = Cc // H
        dev = qkv.device
        lib = _lib()
        s = _stream()
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        QW = Cc if kvc is not None else 3 * Cc
        if kvc is not None:             # compact k|v: its gradient has the compact grid's rows only
            dkvc = torch.empty_like(kvc)
            kv_fine, kv_dfine, kv_off = kvc, dkvc, (0, Cc)
        else:
            dkvc, kv_fine, kv_dfine, kv_off = None, qkv, dqkv, (Cc, 2 * Cc)
        delta = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)

        def grad_target(slot, n_rows):   # where attention bwd writes d(q|k|v)
            if slot in saved:
                t = torch.empty(B, n_rows, Cc, dtype=qkv.dtype, device=dev)
                return t, 0, (n_rows * Cc, Cc, HD)
            return dqkv, slot * Cc, (N * QW, QW, HD)

        tq = grad_target(0, Nq)
        if "kv" in saved:               # dK | dV stacked like the saved pooled pair
            dkv = torch.empty(2, B, Nk, Cc, dtype=qkv.dtype, device=dev)
            tk, tv = (dkv[0], 0, (Nk * Cc, Cc, HD)), (dkv[1], 0, (Nk * Cc, Cc, HD))
        else:
            tk, tv = grad_target(1, Nk), grad_target(2, Nk)
        a = L.AttnArgs()
        a.Q, a.K, a.V = _p(qd[0], qd[1]), _p(kd[0], kd[1]), _p(vd[0], vd[1])
        a.O, a.LSE, a.dO, a.delta = _p(o), _p(lse), _p(do), _p(delta)
        a.dQ, a.dK, a.dV = _p(tq[0], tq[1]), _p(tk[0], tk[1]), _p(tv[0], tv[1])
        a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = _dt(qkv), B, H, Nq, Nk, HD
        a.q_strides = (C.c_int64 * 3)(*qd[2])
        a.k_strides = (C.c_int64 * 3)(*kd[2])
        a.v_strides = (C.c_int64 * 3)(*vd[2])
        a.o_strides = (C.c_int64 * 3)(Nq * Cc, Cc, HD)
        a.do_strides = (C.c_int64 * 3)(Nq * Cc, Cc, HD)
        a.dq_strides = (C.c_int64 * 3)(*tq[2])
        a.dk_strides = (C.c_int64 * 3)(*tk[2])
        a.dv_strides = (C.c_int64 * 3)(*tv[2])
        a.scale = HD ** -0.5
        a.mask_mode, a.mask_T, a.mask_HW = mask_mode, mask_T, mask_HW
        ws = _ws(lib.csts_attn_bwd_workspace(C.byref(a)), dev)
        L.check(lib.csts_attn_bwd(C.byref(a), _p(ws), ws.numel(), s), "csts_attn_bwd")

        grads = {}

        P = ctx.params                      # leaf parameters (wq, gq, bq, wk, gk, bk, wv, gv, bv)

        def pooled_bwd(slot, dy, w, gamma, transposed):
            c, mean, rstd, g = saved[slot]
            pw, pg, pb = P[3 * slot], P[3 * slot + 1], P[3 * slot + 2]
            dc, dg, db = _ln_rows_bwd(dy, c, gamma, mean, rstd, HD, params=(pg, pb))
            defer = _can_defer(pw)
            grouped = defer and _stencil_group_now()
            wsz = (lib.csts_dwconv_wgrad_grouped_workspace if grouped else lib.csts_dwconv_wgrad_workspace)(C.byref(g))
            wws = _ws(wsz, dev)
            dw = torch.empty(HD * 27, dtype=torch.float32, device=dev)
            dwp = None if defer else _p(dw)
            if grouped:
                if transposed:
                    _queue_stencil_wgrad(g, dc, 0, qkv, slot * Cc, wws)
                else:
                    _queue_stencil_wgrad(g, qkv, slot * Cc, dc, 0, wws)
            side = _stencil_side(dc, qkv, wws) if (defer and not grouped and STENCIL_WGRAD_SIDE) else None
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                sw = _stream()
                if grouped:
                    pass
                elif transposed:   # fine = dc (output side), coarse = qkv slot
                    L.check(lib.csts_dwconv_wgrad(C.byref(g), _p(dc), _dt(dc), _p(qkv, slot * Cc), _dt(qkv), dwp, _p(wws),
                                                  wws.numel(), sw), "csts_dwconv_wgrad")
                else:            # fine = qkv slot, coarse = dc
                    L.check(lib.csts_dwconv_wgrad(C.byref(g), _p(qkv, slot * Cc), _dt(qkv), _p(dc), _dt(dc), dwp, _p(wws),
                                                  wws.numel(), sw), "csts_dwconv_wgrad")
            if transposed:
                L.check(lib.csts_dwconv_strided(C.byref(g), _p(dc), _dt(dc), _p(w), _p(dqkv, slot * Cc), _dt(dqkv), s),
                        "csts_dwconv_strided(bwd)")
            else:
                L.check(lib.csts_dwconv_transposed(C.byref(g), _p(dc), _dt(dc), _p(w), _p(dqkv, slot * Cc), _dt(dqkv), s),
                        "csts_dwconv_transposed(bwd)")
            dwv = dw.view(HD, 1, 3, 3, 3)
            if defer:
                _defer(wws, dw, wsz // (HD * 27 * 4), HD * 27)
                _assign_later(pw, dwv)
                dwv = None
            grads[slot] = (dwv, dg, db)

        def pooled_bwd_kv():
            """LayerNorm backward, transposed conv (data gradient) and stencil weight gradient of the k AND v pools,
            one launch each."""
            c2, mean2, rstd2, g = saved["kv"]
            rows = B * Nk * H
            dc2 = torch.empty_like(c2)
            dgb = torch.empty(2, 2 * HD, dtype=torch.float32, device=dev)
            nbytes = lib.csts_layernorm_bwd_workspace(rows, HD)
            lws = _ws(2 * nbytes, dev)
            defer = _can_defer(P[3], P[4], P[5], P[6], P[7], P[8])
            L.check(lib.csts_layernorm_bwd2(_p(dkv), _dt(dkv), _p(c2), _dt(c2), _p(gk), _p(gv), _p(mean2), _p(rstd2), _p(dc2),
                                            _dt(dc2), None if defer else _p(dgb[0]), None if defer else _p(dgb[1]), _p(lws),
                                            lws.numel(), rows, HD, s), "csts_layernorm_bwd2")
            if defer:
                nrow = nbytes // (2 * HD * 4)
                _defer(lws[:nbytes], dgb[0], nrow, 2 * HD)
                _defer(lws[nbytes:], dgb[1], nrow, 2 * HD)
            vp2 = C.c_void_p * 2
            L.check(lib.csts_dwconv_transposed2(C.byref(g), vp2(_p(dc2[0]), _p(dc2[1])), _dt(dc2), vp2(_p(wk), _p(wv)),
                                                vp2(_p(kv_dfine, kv_off[0]), _p(kv_dfine, kv_off[1])), _dt(kv_dfine), s),
                    "csts_dwconv_transposed2")
            grouped = defer and _stencil_group_now()
            wsz = (lib.csts_dwconv_wgrad_grouped_workspace if grouped else lib.csts_dwconv_wgrad_workspace)(C.byref(g))
            wws = _ws(2 * wsz, dev)
            dw = torch.empty(2, HD * 27, dtype=torch.float32, device=dev)
            dwp = vp2(None, None) if defer else vp2(_p(dw[0]), _p(dw[1]))
            if grouped:
                for i in (0, 1):
                    _queue_stencil_wgrad(g, kv_fine, kv_off[i], dc2[i], 0, wws[i * wsz:(i + 1) * wsz])
            side = _stencil_side(dc2, kv_fine, wws) if (defer and not grouped and STENCIL_WGRAD_SIDE) else None
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if not grouped:
                    L.check(lib.csts_dwconv_wgrad2(C.byref(g), vp2(_p(kv_fine, kv_off[0]), _p(kv_fine, kv_off[1])), _dt(kv_fine),
                                               vp2(_p(dc2[0]), _p(dc2[1])), _dt(dc2), dwp, _p(wws), wws.numel(), _stream()), "csts_dwconv_wgrad2")
            if defer:
                nrow = wsz // (HD * 27 * 4)
                _defer(wws[:wsz], dw[0], nrow, HD * 27)
                _defer(wws[wsz:], dw[1], nrow, HD * 27)
            for i, slot in enumerate((1, 2)):
                gw, gg, gb = dw[i].view(HD, 1, 3, 3, 3), dgb[i, :HD], dgb[i, HD:]
                if defer:
                    _assign_later(P[3 * slot], gw)
                    _assign_later(P[3 * slot + 1], gg)
                    _assign_later(P[3 * slot + 2], gb)
                    gw = gg = gb = None
                grads[slot] = (gw, gg, gb)

        if 0 in saved:
            pooled_bwd(0, tq[0], wq, gq, kind == "dec")
        if "kv" in saved:
            pooled_bwd_kv()
        elif 1 in saved:
            pooled_bwd(1, tk[0], wk, gk, False)
            pooled_bwd(2, tv[0], wv, gv, False)
        gq_ = grads.get(0, (None, None, None))
        gk_ = grads.get(1, (None, None, None))
        gv_ = grads.get(2, (None, None, None))
        return (dqkv, dkvc, gq_[0], gq_[1], gq_[2], gk_[0], gk_[1], gk_[2], gv_[0], gv_[1], gv_[2], None)


def attention_inner(qkv, wq, gq, bq, wk, gk, bk, wv, gv, bv, meta, kvc=None):
    return AttnInnerFn.apply(qkv, kvc, wq, gq, bq, wk, gk, bk, wv, gv, bv, meta)


# ----------------------------------------------------------------------------------------- qkv Linear, k|v on the rows that are read
KV_COMPACT_MIN_STRIDE = int(os.environ.get("CSTS_KV_COMPACT", "8"))     # 0 = off; pools with spatial stride >= this take the compact path


def kv_compact_dims(thw, stride_kv):
    """(Hc, Wc) of the compact grid of the token rows a 3x3x3 pool with stride (1, s, s), padding 1 reads (csts_kv_rows_geom):
    3 cells per output cell minus the padding cell in front, minus the last one when it falls outside the grid."""
    out = []
    for n, st in zip(thw[1:], stride_kv[1:]):
        no = (n - 1) // st + 1
        out.append(3 * no - 1 - (1 if (no - 1) * st + 1 > n - 1 else 0))
    return tuple(out)


def kv_compact_ok(thw, stride_kv, has_pool_kv) -> bool:
    return bool(has_pool_kv and KV_COMPACT_MIN_STRIDE > 0 and stride_kv[0] == 1 and stride_kv[1] >= max(3, KV_COMPACT_MIN_STRIDE)
                and stride_kv[2] >= max(3, KV_COMPACT_MIN_STRIDE))


def _kv_rows_geom(B, Cc, thw, stride_kv) -> L.KvRowsGeom:
    g = L.KvRowsGeom()
    g.B, g.C, g.T, g.H, g.W = B, Cc, thw[0], thw[1], thw[2]
    g.sh, g.sw = stride_kv[1], stride_kv[2]
    g.Hc, g.Wc = kv_compact_dims(thw, stride_kv)
    return g


class QkvCompactFn(Function):
    """The qkv Linear of a block whose K/V pools have spatial stride s >= 4 (attention.py:88,130 feeding :104-116): the pools
    are 3x3x3 convs with stride (1, s, s), so only (3/s)^2 of the token rows of k and v are ever read -- 14 % at s = 8, 3.5 % at
    s = 16.  q = x Wq^T + bq over all rows; k|v = x[rows] Wkv^T + bkv over the gathered rows only, laid out on the compact grid
    the pools then walk with stride (1, 3, 3).  Rows of a Linear are independent: the values that ARE computed equal the
    reference's.  Backward: dx = dq Wq + scatter_add(dkv Wkv); dW = [dq^T x ; dkv^T x[rows]]."""

    @staticmethod
    def forward(ctx, x, W, b, thw, stride_kv, out_dt: int, compute: int, w16, w16t):
        _need_gpu(x, W)
        x = x.contiguous()
        B, N, K = x.shape
        Cc = W.shape[0] // 3
        assert W.shape[1] == K and N == thw[0] * thw[1] * thw[2]
        g = _kv_rows_geom(B, K, thw, stride_kv)
        Nc = g.T * g.Hc * g.Wc
        xk = torch.empty(B, Nc, K, dtype=x.dtype, device=x.device)
        L.check(_lib().csts_rows_gather(C.byref(g), _p(x), _dt(x), _p(xk), _stream()), "csts_rows_gather")
        Wop = w16 if (w16 is not None and compute == BF16) else W.contiguous()
        q = torch.empty(B, N, Cc, dtype=torch_dtype(out_dt), device=x.device)
        kv = torch.empty(B, Nc, 2 * Cc, dtype=torch_dtype(out_dt), device=x.device)
        gemm(L.GEMM_NT, x, 0, K, Wop, 0, K, q, Cc, B * N, Cc, K, compute=compute, bias=(b[:Cc] if b is not None else None))
        gemm(L.GEMM_NT, xk, 0, K, Wop, Cc * K, K, kv, 2 * Cc, B * Nc, 2 * Cc, K, compute=compute,
             bias=(b[Cc:] if b is not None else None))
        ctx.save_for_backward(x, xk, Wop)
        ctx.params = (W, b)
        ctx.w16t = w16t if (w16t is not None and compute == BF16 and Wop is w16) else None
        ctx.meta = (B, N, Nc, K, Cc, compute, g)
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv):
        x, xk, W = ctx.saved_tensors
        B, N, Nc, K, Cc, compute, g = ctx.meta
        Wp, bp = ctx.params
        dq, dkv = dq.contiguous(), dkv.contiguous()
        Wt = ctx.w16t
        dx = torch.empty_like(x)
        dxk = torch.empty(B, Nc, K, dtype=torch.float32, device=x.device)     # fp32: one rounding when it joins dx
        if Wt is not None and dq.dtype == L.half_dtype() and USE_W16T:         # NT on the [in][out] twin (columns 0..C | C..3C)
            gemm(L.GEMM_NT, dq, 0, Cc, Wt, 0, 3 * Cc, dx, K, B * N, K, Cc, compute=compute)
            gemm(L.GEMM_NT, dkv, 0, 2 * Cc, Wt, Cc, 3 * Cc, dxk, K, B * Nc, K, 2 * Cc, compute=compute)
        else:
            gemm(L.GEMM_NN, dq, 0, Cc, W, 0, K, dx, K, B * N, K, Cc, compute=compute)
            gemm(L.GEMM_NN, dkv, 0, 2 * Cc, W, Cc * K, K, dxk, K, B * Nc, K, 2 * Cc, compute=compute)
        L.check(_lib().csts_rows_scatter_add(C.byref(g), _p(dxk), _dt(dxk), _p(dx), _dt(dx), _stream()), "csts_rows_scatter_add")
        # weight / bias gradients: rows 0..C from all tokens, rows C..3C from the gathered ones, into ONE gradient tensor
        dW = db = None
        queued = False
        if compute == BF16 and GROUP_WGRADS != "never" and (GROUP_WGRADS != "capture" or torch.cuda.is_current_stream_capturing()) \
                and DEFER_REDUCTIONS and x.dtype == L.half_dtype() and Cc % 8 == 0 and K % 8 == 0 and B * Nc >= 256 \
                and Wp.dtype == torch.float32 and _can_defer(Wp, bp):
            dWf = _grad_buffer(Wp, (3 * Cc, K), x.device)
            dbf = _grad_buffer(bp, (3 * Cc,), x.device) if bp is not None else None
            _wgq.append((dq, x, dWf[:Cc], dbf[:Cc] if dbf is not None else None, B * N, Cc, K))
            _wgq.append((dkv, xk, dWf[Cc:], dbf[Cc:] if dbf is not None else None, B * Nc, 2 * Cc, K))
            _assign_later(Wp, dWf)
            _assign_later(bp, dbf)
            cur = torch.cuda.current_stream()
            _wg_prod[cur.cuda_stream] = cur
            queued = True
        if not queued:
            dW = torch.empty(3 * Cc, K, dtype=torch.float32, device=x.device)
            db = torch.empty(3 * Cc, dtype=torch.float32, device=x.device) if bp is not None else None
            for dy_, x_, r0, nr, tok in ((dq, x, 0, Cc, B * N), (dkv, xk, Cc, 2 * Cc, B * Nc)):
                split = _wgrad_split(nr, K, tok)
                det = split > 1 and _lib().csts_gemm_splitk_workspace(nr, K, tok, split) <= SPLITK_WS_LIMIT
                part = dW[r0:r0 + nr]
                if split > 1 and not det:
                    part.zero_()
                cs = gemm(L.GEMM_TN, dy_, 0, nr, x_, 0, K, part, K, nr, K, tok, compute=compute, split_k=split, deterministic=det,
                          want_colsum=db is not None)
                if db is not None:
                    db[r0:r0 + nr].copy_(cs if cs is not None else colsum(dy_, 1, tok, nr))
            dW = dW.to(Wp.dtype)
        return dx, dW, db, None, None, None, None, None, None


def qkv_compact(x, W, b, thw, stride_kv, *, out_dt, compute, w16=None, w16t=None):
    """(q (B, N, C), k|v (B, Nc, 2C) on the compact grid) -- see QkvCompactFn."""
    return QkvCompactFn.apply(x, W, b, list(thw), tuple(stride_kv), out_dt, compute, w16, w16t)


def attention_probs(qkv, B, N, Cc, H, lse, mask_mode, mask_T, mask_HW):
    """Probabilities of an un-pooled attention (fusion blocks) for return_spatial_attn / return_temporal_attn
    (av_attention.py:150-153,358-359); forward-only visualisation output."""
    HD = Cc // H
    probs = torch.empty(B, H, N, N, dtype=torch.float32, device=qkv.device)
    a = L.AttnArgs()
    a.Q, a.K, a.V = _p(qkv, 0), _p(qkv, Cc), _p(qkv, 2 * Cc)
    a.LSE = _p(lse)
    a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = _dt(qkv), B, H, N, N, HD
    st = (C.c_int64 * 3)(N * 3 * Cc, 3 * Cc, HD)
    a.q_strides = st
    a.k_strides = st
    a.v_strides = st
    a.scale = HD ** -0.5
    a.mask_mode, a.mask_T, a.mask_HW = mask_mode, mask_T, mask_HW
    L.check(_lib().csts_attn_probs(C.byref(a), _p(probs), _stream()), "csts_attn_probs")
    return probs


class AudioAttnFn(Function):
    """Per-token weight from the audio -> pixel attention of the spatial fusion block (MVIT.SPATIAL_AUDIO_ATTN):
    av_attention.py:356-370 (probabilities of audio token t over frame t's video tokens, min-max rescaled per head) and the
    head mean of custom_multimodal_builder.py:438.  Differentiable w.r.t. the block's qkv projection (the reference
    back-propagates through the rescaled map in train mode).  Returns (wmap (B, T*HW), audio_attn (B, H, T, HW))."""

    @staticmethod
    def forward(ctx, qkv, T: int, HW: int, Cc: int, H: int):
        _need_gpu(qkv)
        qkv = qkv.contiguous()
        B = qkv.shape[0]
        if qkv.shape[1] != T * HW + T or qkv.shape[2] != 3 * Cc:
            raise L.CstsError(f"audio_attn: qkv {tuple(qkv.shape)} is not (B, {T * HW + T}, {3 * Cc})")
        aa = torch.empty(B, H, T, HW, dtype=torch.float32, device=qkv.device)
        wmap = torch.empty(B, T * HW, dtype=torch.float32, device=qkv.device)
        scale = (Cc // H) ** -0.5
        L.check(_lib().csts_audio_attn_fwd(_p(qkv), _dt(qkv), _p(aa), _p(wmap), B, T, HW, Cc, H, scale, _stream()),
                "csts_audio_attn_fwd")
        ctx.save_for_backward(qkv)
        ctx.meta = (B, T, HW, Cc, H, scale)
        ctx.mark_non_differentiable(aa)
        ctx.set_materialize_grads(False)
        return wmap, aa

    @staticmethod
    def backward(ctx, dwmap, _daa):
        (qkv,) = ctx.saved_tensors
        B, T, HW, Cc, H, scale = ctx.meta
        if dwmap is None:
            return None, None, None, None, None
        dwmap = dwmap.contiguous().float()
        dqkv = torch.zeros_like(qkv)
        L.check(_lib().csts_audio_attn_bwd(_p(qkv), _dt(qkv), _p(dwmap), _p(dqkv), B, T, HW, Cc, H, scale, _stream()),
                "csts_audio_attn_bwd")
        return dqkv, None, None, None, None


def audio_attn(qkv, T, HW, Cc, H):
    return AudioAttnFn.apply(qkv, int(T), int(HW), int(Cc), int(H))


class RowWeightFn(Function):
    """x (B, N, C) * w (B, N)[:, :, None]  (x_temporal * audio_attn, custom_multimodal_builder.py:439-440)."""

    @staticmethod
    def forward(ctx, x, w):
        _need_gpu(x, w)
        x = x.contiguous()
        w = w.contiguous().float()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        if w.numel() != M:
            raise L.CstsError("row_weight: one weight per row expected")
        y = torch.empty_like(x)
        L.check(_lib().csts_scale_rows(_p(x), _dt(x), _p(w), 1, _p(y), _dt(y), M, Cc, _stream()), "csts_scale_rows")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        Cc = x.shape[-1]
        M = x.numel() // Cc
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            L.check(_lib().csts_scale_rows(_p(dy), _dt(dy), _p(w), 1, _p(dx), _dt(dx), M, Cc, _stream()), "csts_scale_rows(bwd)")
        if ctx.needs_input_grad[1]:
            dw = torch.empty(w.shape, dtype=torch.float32, device=x.device)
            L.check(_lib().csts_rowdot2(_p(dy), _dt(dy), _p(x), _dt(x), _p(dw), M, Cc, _stream()), "csts_rowdot2")
        return dx, dw


def row_weight(x, w):
    return RowWeightFn.apply(x, w)


# ----------------------------------------------------------------------------------------- resampling
def _pool_geom(B, Cc, thw_in, thw_out, stride) -> L.PoolGeom:
    g = L.PoolGeom()
    g.B, g.C = B, Cc
    g.Ti, g.Hi, g.Wi = thw_in
    g.To, g.Ho, g.Wo = thw_out
    g.st, g.sh, g.sw = stride
    return g


class MaxPoolFn(Function):
    """pool_skip MaxPool3d on tokens (attention.py:193-195,234-236,240)."""

    @staticmethod
    def forward(ctx, x, thw, stride):
        _need_gpu(x)
        x = x.contiguous()
        B, N, Cc = x.shape
        k = [s + 1 if s > 1 else 1 for s in stride]
        out = [(d + 2 * (kk // 2) - kk) // s + 1 for d, kk, s in zip(thw, k, stride)]
        g = _pool_geom(B, Cc, thw, out, stride)
        y = torch.empty(B, out[0] * out[1] * out[2], Cc, dtype=x.dtype, device=x.device)
        arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        L.check(_lib().csts_maxpool_fwd(C.byref(g), _p(x), _dt(x), _p(y), _p(arg), _stream()), "csts_maxpool_fwd")
        ctx.save_for_backward(arg)
        ctx.g = g
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        L.check(_lib().csts_maxpool_bwd(C.byref(ctx.g), _p(dy), _dt(dy), _p(arg), _p(dx), _stream()), "csts_maxpool_bwd")
        return dx, None, None


class TrilinearFn(Function):
    """nn.Upsample(scale_factor=stride, mode='trilinear') on tokens (attention.py:463-467,471); with `addend`
    it is also feat + F.interpolate(en_feat) of custom_multimodal_builder.py:479."""

    @staticmethod
    def forward(ctx, x, thw, stride, addend):
        _need_gpu(x)
        x = x.contiguous()
        B, N, Cc = x.shape
        out = [d * s for d, s in zip(thw, stride)]
        g = _pool_geom(B, Cc, thw, out, stride)
        y = torch.empty(B, out[0] * out[1] * out[2], Cc, dtype=x.dtype, device=x.device)
        if addend is not None:
            addend = addend.contiguous()
        L.check(_lib().csts_trilinear_fwd(C.byref(g), _p(x), _dt(x), _p(addend), _dt(addend) if addend is not None else 0,
                                          _p(y), _dt(y), _stream()), "csts_trilinear_fwd")
        ctx.g = g
        ctx.shape = x.shape
        ctx.has_add = addend is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        L.check(_lib().csts_trilinear_bwd(C.byref(ctx.g), _p(dy), _dt(dy), _p(dx), _dt(dx), _stream()), "csts_trilinear_bwd")
        return dx, None, None, (dy if ctx.has_add else None)


def maxpool_skip(x, thw, stride):
    return MaxPoolFn.apply(x, list(thw), list(stride))


def trilinear(x, thw, stride, addend=None):
    return TrilinearFn.apply(x, list(thw), list(stride), addend)


# ----------------------------------------------------------------------------------------- patch embed
class PatchEmbedFn(Function):
    """PatchEmbed.forward (stem_helper.py:35-38) + separable pos-embed add (custom_multimodal_builder.py:362-370):
    im2col -> MFMA GEMM with bias and the (broadcast) positional embedding in the epilogue."""

    @staticmethod
    def forward(ctx, x, W, b, pos_s, pos_t, kernel, stride, padding, act_dt: int, compute: int):
        _need_gpu(x, W)
        x = x.contiguous()
        B, Cin, T, H, Wd = x.shape
        Cout = W.shape[0]
        K = Cin * kernel[0] * kernel[1] * kernel[2]
        Kpad = (K + 31) // 32 * 32
        g = L.Im2colGeom()
        g.B, g.Cin, g.T, g.H, g.W = B, Cin, T, H, Wd
        g.kernel = (C.c_int * 3)(*kernel)
        g.stride = (C.c_int * 3)(*stride)
        g.padding = (C.c_int * 3)(*padding)
        out = [(d + 2 * p - k) // s + 1 for d, p, k, s in zip((T, H, Wd), padding, kernel, stride)]
        g.To, g.Ho, g.Wo = out
        g.Kpad = Kpad
        N = out[0] * out[1] * out[2]
        M = B * N
        col = torch.empty(M, Kpad, dtype=torch_dtype(act_dt), device=x.device)
        L.check(_lib().csts_im2col(C.byref(g), _p(x), _dt(x), _p(col), act_dt, _stream()), "csts_im2col")
        Wp = torch.zeros(Cout, Kpad, dtype=torch.float32, device=x.device)
        Wp[:, :K] = W.reshape(Cout, K)            # weight-layout plumbing (42 K elements)
        pos = torch.empty(N, Cout, dtype=torch.float32, device=x.device)
        HW = out[1] * out[2]
        L.check(_lib().csts_posembed_build(_p(pos_s), _p(pos_t), _p(pos), out[0], HW, Cout, _stream()), "csts_posembed_build")
        y = torch.empty(B, N, Cout, dtype=torch.float32, device=x.device)
        gemm(L.GEMM_NT, col, 0, Kpad, Wp, 0, Kpad, y, Cout, M, Cout, Kpad, compute=compute, bias=b, residual=pos, ldr=Cout,
             res_row_mod=N)
        ctx.save_for_backward(col)
        ctx.meta = (B, N, out[0], HW, Cout, K, Kpad, compute, tuple(W.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        (col,) = ctx.saved_tensors
        B, N, T, HW, Cout, K, Kpad, compute, wshape = ctx.meta
        dy = dy.contiguous()
        M = B * N
        dWp = _wgrad(dy, col, M, Cout, Kpad, compute)
        dW = dWp[:, :K].reshape(wshape)
        dpos = colsum(dy, 1, B, N * Cout)                       # sum over batch -> (N*Cout): the only pass over dy itself
        dps = colsum(dpos, 1, T, HW * Cout).view(1, HW, Cout)   # sum over t
        dpt = colsum(dpos, T, HW, Cout).view(1, T, Cout)        # sum over hw, per t
        db = colsum(dpt, 1, T, Cout)                            # bias gradient = sum over every token: from the T partial rows
        return None, dW, db, dps, dpt, None, None, None, None, None


def patch_embed(x, W, b, pos_s, pos_t, kernel, stride, padding, act_dt, compute):
    return PatchEmbedFn.apply(x, W, b, pos_s, pos_t, list(kernel), list(stride), list(padding), act_dt, compute)


# ----------------------------------------------------------------------------------------- fusion conv
# Data-parallel factor exchange (csts_amd.train.SegmentedTrainStep, CSTS_AMD.FUSION_GRAD_FACTORS): the weight gradient of a fusion
# conv is dW = dY^T A with only B*T' token rows (32 at b = 4, 16 frames) -- a rank-32 update of a 768 x 49152 matrix, 151 MB in
# fp32, 60 % of all gradient bytes for the three of them.  Across W ranks the averaged gradient is (1/W) [dY_0; ..; dY_W-1]^T
# [A_0; ..; A_W-1]: the ranks exchange the FACTORS (all-gather of 32 x 768 fp32 + 32 x 49152 16-bit rows: 3.2 MB per conv) and each
# forms the product itself, instead of all-reducing the 151 MB.  While a sink is set, FusionConvFn.backward hands its factors over
# and computes no dW (the parameter's gradient is written later by the step that owns the sink).
_factor_sink = None          # None, or {"params": set of weight data_ptrs, "items": [(W, dy, A, compute)], optional "accept": f(W, rows)}


def set_factor_sink(sink=None):
    global _factor_sink
    _factor_sink = sink


class FusionConvFn(Function):
    """Conv3d(C, C, kernel (1,8,8)) over the folded token grid -> (B, T, C)
    (custom_multimodal_builder.py:227-229,420-421,442-445): token fold (batched transpose) + skinny split-K GEMM
    that streams the 37.7 M-parameter weight once, in its native (Cout, Cin*H*W) layout."""

    @staticmethod
    def forward(ctx, x, W, b, T: int, HW: int, act_dt: int, compute: int, w16=None):
        _need_gpu(x, W)
        x = x.contiguous()
        B, N, Cc = x.shape
        Cout = W.shape[0]
        BT = B * T
        K = Cc * HW
        A = torch.empty(BT, K, dtype=torch_dtype(act_dt), device=x.device)
        L.check(_lib().csts_transpose_batched(_p(x), _dt(x), _p(A), act_dt, BT, HW, Cc, _stream()), "csts_transpose_batched")
        y = torch.empty(B, T, Cout, dtype=torch.float32, device=x.device)
        Wop = w16 if (w16 is not None and compute == BF16 and w16.numel() == W.numel()) else W     # bf16 shadow: half the stream
        Wv = Wop.reshape(Cout, K)
        split = max(1, min(128, K // 256))
        gemm(L.GEMM_NT, A, 0, K, Wv, 0, K, y, Cout, BT, Cout, K, compute=compute, bias=b, split_k=split)
        ctx.save_for_backward(A, W, Wop)
        ctx.meta = (B, T, HW, Cc, Cout, compute, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        A, W, Wop = ctx.saved_tensors
        B, T, HW, Cc, Cout, compute, xdtype = ctx.meta
        BT, K = B * T, Cc * HW
        dy = dy.contiguous()
        Wv = Wop.reshape(Cout, K)
        sink = _factor_sink
        if (sink is not None and W.data_ptr() in sink["params"] and ctx.needs_input_grad[1]
                and ("accept" not in sink or sink["accept"](W, BT))):
            sink["items"].append((W, dy, A, compute))      # factors only: the owner of the sink forms dW after the exchange
            dW = None
        else:
            dW = _grad_buffer(W, (Cout, K), dy.device)        # the data-parallel chain: straight into its all-reduce bucket
            gemm(L.GEMM_TN, dy, 0, Cout, A, 0, K, dW, K, Cout, K, BT, compute=compute)
        db = colsum(dy, 1, BT, Cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty_like(A)
            gemm(L.GEMM_NN, dy, 0, Cout, Wv, 0, K, dA, K, BT, K, Cout, compute=compute)
            dx = torch.empty(B, T * HW, Cc, dtype=xdtype, device=dy.device)
            L.check(_lib().csts_transpose_batched(_p(dA), _dt(dA), _p(dx), _dt(dx), BT, Cc, HW, _stream()),
                    "csts_transpose_batched(bwd)")
        return dx, (dW.view(W.shape) if dW is not None else None), db, None, None, None, None, None


def fusion_conv(x, W, b, T, HW, act_dt, compute, w16=None):
    return FusionConvFn.apply(x, W, b, T, HW, act_dt, compute, w16)


# ----------------------------------------------------------------------------------------- glue
class ReweightFn(Function):
    """x.view(B,T,H,W,C) * w[:, :, None, None, :]  (custom_multimodal_builder.py:454-461)."""

    @staticmethod
    def forward(ctx, x, w, T: int, HW: int):
        _need_gpu(x, w)
        x = x.contiguous().float()
        w = w.contiguous().float()
        B, N, Cc = x.shape
        y = torch.empty_like(x)
        L.check(_lib().csts_reweight_fwd(_p(x), _p(w), _p(y), B * T, HW, Cc, _stream()), "csts_reweight_fwd")
        ctx.save_for_backward(x, w)
        ctx.meta = (B * T, HW, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        BT, HW, Cc = ctx.meta
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        L.check(_lib().csts_reweight_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), BT, HW, Cc, _stream()), "csts_reweight_bwd")
        return dx, dw, None, None


class TokenMeanFn(Function):
    """x.mean(dim=1)  (custom_multimodal_builder.py:493-494)."""

    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = x.contiguous().float()
        B, N, Cc = x.shape
        out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        L.check(_lib().csts_token_mean_fwd(_p(x), _p(out), B, N, Cc, _stream()), "csts_token_mean_fwd")
        ctx.meta = (B, N, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, Cc = ctx.meta
        dout = dout.contiguous().float()
        dx = torch.empty(B, N, Cc, dtype=torch.float32, device=dout.device)
        L.check(_lib().csts_token_mean_bwd(_p(dout), _p(dx), B, N, Cc, _stream()), "csts_token_mean_bwd")
        return dx


class AddFn(Function):
    """a + b (decoder encoder-skip adds, custom_multimodal_builder.py:467,470,473)."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        a = a.contiguous()
        b = b.contiguous()
        out = torch.empty_like(a)
        L.check(_lib().csts_axpby(_p(a), _dt(a), _p(b), _dt(b), _p(out), _dt(out), a.numel(), 1.0, 1.0, _stream()), "csts_axpby")
        ctx.bdtype = b.dtype
        return out

    @staticmethod
    def backward(ctx, d):
        return d, (d if d.dtype == ctx.bdtype else d.to(ctx.bdtype))


class TapFn(Function):
    """Identity with two outputs for a residual-stream tensor that has TWO consumers (the encoder features the decoder
    skips re-use, custom_multimodal_builder.py:384-403,467-479): both gradients meet in ONE backward kernel
    (d = d_a + d_b, plus the bf16 copy the next block's GEMMs read) instead of autograd's own in-place accumulation."""

    @staticmethod
    def forward(ctx, x, compute: int):
        _need_gpu(x)
        ctx.compute = compute
        ctx.copy_scale = getattr(x, "_csts_prod_scale", None)     # x = f(...) * s + residual: its gradient's next reader wants s * dx
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, da, db):
        if da is None or db is None:
            return (da if da is not None else db), None
        da, db = da.contiguous(), db.contiguous()
        out = torch.empty(da.shape, dtype=torch.float32, device=da.device)
        want16 = ctx.compute == BF16 and BF16_GRAD_COPY
        out16 = torch.empty(da.shape, dtype=L.half_dtype(), device=da.device) if want16 else None
        cs = ctx.copy_scale if out16 is not None else None
        if cs is not None:
            rows = da.numel() // da.shape[-1]
            if rows % cs[1] != 0 or cs[0].numel() * cs[1] != rows or (cs[1] * da.shape[-1]) % 4 != 0:
                cs = None
        L.check(_lib().csts_add2_scaled_copy(_p(da), _dt(da), _p(db), _dt(db), _p(out), _p(out16), _p(cs[0]) if cs else None,
                                             cs[1] * da.shape[-1] if cs else 4, da.numel(), _stream()), "csts_add2")
        if out16 is not None:
            _attach16(out, out16, cs)
        return out, None


# ----------------------------------------------------------------------------------------- token-axis cat / split
# torch.cat(dim=1) and x[:, :n] / x[:, n:] of (B, N, C) tensors as ONE library launch per direction (csts_copy_token_segments).  Through
# torch the fusion head's joins and cuts were ~30 cat / slice-copy / zero-fill / gradient-add nodes per captured step (0.15 ms of
# 5 us kernels).  CSTS_TOKEN_GLUE=0: the torch ops again (same values: pure copies).
TOKEN_GLUE = os.environ.get("CSTS_TOKEN_GLUE", "1") != "0"


def _copy_segments(pairs, B, Cc, dt):
    """pairs: [(src tensor, src batch stride, src element offset, dst tensor, dst batch stride, dst element offset, rows)]"""
    arr = (L.TokenSegment * len(pairs))()
    for i, (src, sbs, soff, dst, dbs, doff, n) in enumerate(pairs):
        arr[i].src, arr[i].dst = src.data_ptr(), dst.data_ptr()
        arr[i].src_bs, arr[i].dst_bs, arr[i].src_off, arr[i].dst_off, arr[i].n = sbs, dbs, soff, doff, n
    L.check(_lib().csts_copy_token_segments(arr, len(pairs), B, Cc, dt, _stream()), "csts_copy_token_segments")


class CatTokensFn(Function):
    """torch.cat([a, b], dim=1) of (B, Na, C) and (B, Nb, C)."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        a, b = a.contiguous(), b.contiguous()
        B, Na, Cc = a.shape
        Nb = b.shape[1]
        out = torch.empty(B, Na + Nb, Cc, dtype=a.dtype, device=a.device)
        _copy_segments([(a, Na * Cc, 0, out, (Na + Nb) * Cc, 0, Na), (b, Nb * Cc, 0, out, (Na + Nb) * Cc, Na * Cc, Nb)], B, Cc, _dt(a))
        ctx.meta = (B, Na, Nb, Cc)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Na, Nb, Cc = ctx.meta
        dout = dout.contiguous()
        da = torch.empty(B, Na, Cc, dtype=dout.dtype, device=dout.device)
        db = torch.empty(B, Nb, Cc, dtype=dout.dtype, device=dout.device)
        _copy_segments([(dout, (Na + Nb) * Cc, 0, da, Na * Cc, 0, Na), (dout, (Na + Nb) * Cc, Na * Cc, db, Nb * Cc, 0, Nb)], B, Cc, _dt(dout))
        return da, db


class SplitTokensFn(Function):
    """(x[:, :n], x[:, n:]) of a (B, N, C) tensor as two contiguous tensors."""

    @staticmethod
    def forward(ctx, x, n: int):
        _need_gpu(x)
        x = x.contiguous()
        B, N, Cc = x.shape
        a = torch.empty(B, n, Cc, dtype=x.dtype, device=x.device)
        b = torch.empty(B, N - n, Cc, dtype=x.dtype, device=x.device)
        _copy_segments([(x, N * Cc, 0, a, n * Cc, 0, n), (x, N * Cc, n * Cc, b, (N - n) * Cc, 0, N - n)], B, Cc, _dt(x))
        ctx.meta = (B, N, n, Cc)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        B, N, n, Cc = ctx.meta
        # a part nobody used has no gradient (None): zeros, like the slice's backward
        ref = da if da is not None else db
        if da is None:
            da = torch.zeros(B, n, Cc, dtype=ref.dtype, device=ref.device)
        if db is None:
            db = torch.zeros(B, N - n, Cc, dtype=ref.dtype, device=ref.device)
        da, db = da.contiguous(), db.contiguous()
        dx = torch.empty(B, N, Cc, dtype=da.dtype, device=da.device)
        _copy_segments([(da, n * Cc, 0, dx, N * Cc, 0, n), (db, (N - n) * Cc, 0, dx, N * Cc, n * Cc, N - n)], B, Cc, _dt(da))
        return dx, None


def cat_tokens(a, b):
    if not TOKEN_GLUE or a.dtype != b.dtype or (a.shape[2] * a.element_size()) % 16:
        return torch.cat([a, b], dim=1)
    return CatTokensFn.apply(a, b)


def split_tokens(x, n: int):
    if not TOKEN_GLUE or (x.shape[2] * x.element_size()) % 16:
        return x[:, :n, :], x[:, n:, :]
    return SplitTokensFn.apply(x, n)


def tap(x, compute):
    """(x_a, x_b): two aliases of x, one per consumer."""
    return TapFn.apply(x, compute)


def cast(x: torch.Tensor, dt: int) -> torch.Tensor:
    out = torch.empty(x.shape, dtype=torch_dtype(dt), device=x.device)
    x = x.contiguous()
    L.check(_lib().csts_axpby(_p(x), _dt(x), None, 0, _p(out), dt, x.numel(), 1.0, 0.0, _stream()), "csts_axpby(cast)")
    return out


def cast_into(dst: torch.Tensor, src: torch.Tensor, scale: float = 1.0):
    """dst = scale * src across dtypes (flat gradient buckets: fp32 -> the 16-bit type before the all-reduce).  GPU: one
    vectorised csts_axpby launch; CPU tensors (the gloo tests of the bucket logic): torch."""
    assert dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    if not dst.is_cuda:
        dst.copy_(src if scale == 1.0 else src * scale)
        return dst
    L.check(_lib().csts_axpby(_p(src), _dt(src), None, 0, _p(dst), _dt(dst), src.numel(), float(scale), 0.0, _stream()), "csts_axpby(cast)")
    return dst


def reweight(x, w, T, HW):
    return ReweightFn.apply(x, w, T, HW)


def token_mean(x):
    return TokenMeanFn.apply(x)


def add(a, b):
    return AddFn.apply(a, b)


# ----------------------------------------------------------------------------------------- head + losses
class ClassifierFn(Function):
    """classifier(feat + F.interpolate(en_feat, (2T', H, W), 'trilinear'))  with classifier = Conv3d(96, 1, 1)
    (custom_multimodal_builder.py:476-481)."""

    @staticmethod
    def forward(ctx, feat, en_feat, w, b, thw_en, compute: int = F32):
        _need_gpu(feat, en_feat)
        ctx.compute = compute
        feat = feat.contiguous()
        en_feat = en_feat.contiguous()
        B, N2, Cc = feat.shape
        out_thw = [thw_en[0] * 2, thw_en[1], thw_en[2]]
        g = _pool_geom(B, Cc, thw_en, out_thw, (2, 1, 1))
        z = torch.empty_like(feat)
        L.check(_lib().csts_trilinear_fwd(C.byref(g), _p(en_feat), _dt(en_feat), _p(feat), _dt(feat), _p(z), _dt(z), _stream()),
                "csts_trilinear_fwd(head)")
        logits = torch.empty(B * N2, dtype=torch.float32, device=feat.device)
        wf = w.reshape(-1).contiguous()
        L.check(_lib().csts_rowdot_fwd(_p(z), _dt(z), _p(wf), _p(b), _p(logits), B * N2, Cc, _stream()), "csts_rowdot_fwd")
        ctx.save_for_backward(z, wf)
        ctx.g = g
        ctx.meta = (B, N2, Cc, tuple(w.shape), en_feat.shape, en_feat.dtype)
        return logits.view(B, 1, out_thw[0], out_thw[1], out_thw[2])

    @staticmethod
    def backward(ctx, dl):
        z, wf = ctx.saved_tensors
        B, N2, Cc, wshape, enshape, endtype = ctx.meta
        dl = dl.contiguous().view(-1).float()
        M = B * N2
        dz = torch.empty_like(z)
        L.check(_lib().csts_rowdot_dx(_p(dl), _p(wf), _p(dz), _dt(dz), M, Cc, _stream()), "csts_rowdot_dx")
        if ctx.compute == BF16 and BF16_GRAD_COPY and dz.dtype == torch.float32:
            # the last decoder block has no drop-path and nothing else would give its GEMMs a bf16 copy of this gradient:
            # without one its two data-gradient GEMMs and its weight gradients stream the fp32 tensor (M = 262 k rows)
            dz16 = torch.empty(dz.shape, dtype=L.half_dtype(), device=dz.device)
            L.check(_lib().csts_rowdot_dx(_p(dl), _p(wf), _p(dz16), BF16, M, Cc, _stream()), "csts_rowdot_dx(bf16)")
            _attach16(dz, dz16)
        dw = colsum(z, 1, M, Cc, row_weight=dl).view(wshape)
        db = colsum(dl, 1, M, 1)
        den = torch.empty(enshape, dtype=endtype, device=dl.device)
        L.check(_lib().csts_trilinear_bwd(C.byref(ctx.g), _p(dz), _dt(dz), _p(den), _dt(den), _stream()), "csts_trilinear_bwd(head)")
        return dz, den, dw, db, None, None


def classifier_head(feat, en_feat, w, b, thw_en, compute: int = F32):
    return ClassifierFn.apply(feat, en_feat, w, b, list(thw_en), compute)


class FrameSoftmaxFn(Function):
    """frame_softmax(logits, temperature): softmax over H*W per (b, t)  (slowfast/utils/utils.py:5-12)."""

    @staticmethod
    def forward(ctx, logits, temperature: float):
        _need_gpu(logits)
        x = logits.contiguous().float()
        n = x.shape[-1] * x.shape[-2]
        rows = x.numel() // n
        p = torch.empty_like(x)
        L.check(_lib().csts_softmax_fwd(_p(x), _p(p), rows, n, temperature, _stream()), "csts_softmax_fwd")
        ctx.save_for_backward(x)
        ctx.meta = (rows, n, temperature)
        return p

    @staticmethod
    def backward(ctx, dp):
        (x,) = ctx.saved_tensors
        rows, n, temperature = ctx.meta
        dp = dp.contiguous().float()
        dx = torch.empty_like(x)
        L.check(_lib().csts_softmax_bwd(_p(x), _p(dp), _p(dx), rows, n, temperature, _stream()), "csts_softmax_bwd")
        return dx, None


class KLDivFn(Function):
    """KLDiv.forward (slowfast/models/losses.py:59-82): sum_t KL(p || q) / (T log HW), mean over batch."""

    @staticmethod
    def forward(ctx, pred, target):
        _need_gpu(pred)
        p = pred.contiguous().float()
        B, T, H, W = p.shape[0], p.shape[2], p.shape[3], p.shape[4]
        n = H * W
        rows = p.numel() // n
        q = target.contiguous().float() if target is not None else None
        scale = 1.0 / (T * math.log(n) * B)
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        ws = torch.empty(rows, dtype=torch.float32, device=p.device)
        L.check(_lib().csts_kldiv_fwd(_p(p), _p(q), _p(loss), _p(ws), rows, n, scale, _stream()), "csts_kldiv_fwd")
        if q is None:   # uniform prior: - log(1/HW) per frame (losses.py:69-73)
            loss = loss + math.log(n) * T * B * scale
        ctx.save_for_backward(p, q)
        ctx.meta = (rows, n, scale)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        p, q = ctx.saved_tensors
        rows, n, scale = ctx.meta
        g = g.contiguous().float().view(1)
        dp = torch.empty_like(p)
        L.check(_lib().csts_kldiv_bwd(_p(p), _p(q), _p(g), _p(dp), rows, n, scale, _stream()), "csts_kldiv_bwd")
        return dp, None


class RowNormFn(Function):
    """a / max(||a||, eps) per row (sim_matrix, slowfast/utils/utils.py:20-22)."""

    @staticmethod
    def forward(ctx, a, eps: float):
        _need_gpu(a)
        a = a.contiguous().float()
        rows, D = a.shape
        an = torch.empty_like(a)
        nrm = torch.empty(rows, dtype=torch.float32, device=a.device)
        L.check(_lib().csts_rownorm_fwd(_p(a), _p(an), _p(nrm), rows, D, eps, _stream()), "csts_rownorm_fwd")
        ctx.save_for_backward(a, nrm)
        ctx.eps = eps
        return an

    @staticmethod
    def backward(ctx, dan):
        a, nrm = ctx.saved_tensors
        dan = dan.contiguous().float()
        da = torch.empty_like(a)
        L.check(_lib().csts_rownorm_bwd(_p(a), _p(nrm), _p(dan), _p(da), a.shape[0], a.shape[1], ctx.eps, _stream()),
                "csts_rownorm_bwd")
        return da, None


class EgoNCEFn(Function):
    """EgoNCE.forward (slowfast/models/losses.py:157-170), temperature 0.05."""

    @staticmethod
    def forward(ctx, sim, temperature: float):
        _need_gpu(sim)
        x = sim.contiguous().float()
        n = x.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        lr = torch.empty(n, dtype=torch.float32, device=x.device)
        lc = torch.empty_like(lr)
        L.check(_lib().csts_egonce_fwd(_p(x), _p(loss), _p(lr), _p(lc), n, temperature, _stream()), "csts_egonce_fwd")
        ctx.save_for_backward(x, lr, lc)
        ctx.meta = (n, temperature)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        x, lr, lc = ctx.saved_tensors
        n, temperature = ctx.meta
        g = g.contiguous().float().view(1)
        dx = torch.empty_like(x)
        L.check(_lib().csts_egonce_bwd(_p(x), _p(lr), _p(lc), _p(g), _p(dx), n, temperature, _stream()), "csts_egonce_bwd")
        return dx, None


def frame_softmax(logits, temperature):
    return FrameSoftmaxFn.apply(logits, float(temperature))


def kldiv(pred, target=None):
    return KLDivFn.apply(pred, target)


def sim_matrix(a, b, eps=1e-8):
    an = RowNormFn.apply(a, eps)
    bn = RowNormFn.apply(b, eps)
    return linear(an, bn, None, out_dt=F32, compute=F32)


def egonce(sim, temperature=0.05):
    return EgoNCEFn.apply(sim, float(temperature))
