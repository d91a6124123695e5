"""Device-side optimizer step for the CSTS iteration: L2 gradient clip + AdamW + bf16 shadow
refresh in three kernel launches over the whole parameter set (csts_adamw_step), instead of torch's ~65
multi-tensor launches (fused AdamW + _foreach norm + _foreach mul + shadow copies).

Mirrors the reference recipe: slowfast/models/optimizer.py:11-108 (AdamW, eps 1e-8, weight decay 0 for 1-D
parameters and biases), tools/train_avgaze_net.py:101-109 (unscale -> clip_grad_norm_(1.0) -> step) and the
per-iteration learning rate of slowfast/models/optimizer.py:122-130 (set through ``param_groups[i]["lr"]``).
The interface follows torch.optim.Optimizer where the reference touches it: param_groups, step(), zero_grad(),
state_dict() / load_state_dict().
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import os

import torch

from . import lib as L

# elements per workgroup of the clip / AdamW launches.  Round 5: 16384 (64 KiB of fp32 gradient, ~3400 workgroups for the trunk parameters) instead of
# 65536 (~1100): clip + AdamW 1.245 -> 1.165 ms, step -0.07 ms (8192 / 4096 alike, 131072 alike, 262144 +0.17 ms; gpurun_out/r5as, r5at).  A/B: CSTS_OPT_CHUNK
CHUNK = int(os.environ.get("CSTS_OPT_CHUNK", "16384"))


class FusedAdamW:
    def __init__(self, param_groups: List[Dict], lr: float, betas=(0.9, 0.999), eps: float = 1e-8, max_grad_norm: float = 0.0,
                 shadows: Dict[int, torch.Tensor] = None, loss_scaling: bool = False, init_scale: float = 65536.0,
                 growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        """param_groups: [{"params": [...], "weight_decay": wd}, ...]; shadows: id(param) -> bf16 tensor kept equal
        to the parameter (the GEMMs' bf16 operand)."""
        self.param_groups = [dict(g) for g in param_groups]
        params = [p for g in self.param_groups for p in g["params"]]
        assert params and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params), \
            "FusedAdamW runs on MI355X: fp32 contiguous GPU parameters"
        self.device = params[0].device
        self.params = params
        self.betas, self.eps, self.max_grad_norm = betas, float(eps), float(max_grad_norm)
        self._lr = torch.tensor(float(lr), dtype=torch.float32, device=self.device)
        for g in self.param_groups:
            g["lr"] = self._lr           # one device scalar shared by all groups (the reference sets them all alike)
        shadows = shadows or {}
        n_total = sum(p.numel() for p in params)
        # first / second moments: one flat buffer each, sliced per tensor (16-byte aligned slices)
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.exp_avg = torch.zeros(o, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(o, dtype=torch.float32, device=self.device)
        self._m = [self.exp_avg[a:a + p.numel()] for a, p in zip(offs, params)]
        self._v = [self.exp_avg_sq[a:a + p.numel()] for a, p in zip(offs, params)]
        self.state_t = torch.zeros(4, dtype=torch.float32, device=self.device)     # step, grad norm, clip coefficient, skipped
        # dynamic loss scaling (torch.cuda.amp.GradScaler defaults; tools/train_avgaze_net.py:277): {scale, growth tracker} on the
        # device -- the training harness multiplies the loss by loss_scale before backward, the kernels unscale, detect non-finite
        # gradients, skip the step and update the scale (scaler.unscale_ / step / update, train_avgaze_net.py:101-109)
        self.scaler_t = torch.tensor([float(init_scale), 0.0], dtype=torch.float32, device=self.device) if loss_scaling else None
        self.scaler_cfg = (float(growth_factor), float(backoff_factor), int(growth_interval))
        self.grad_dt = L.F32          # L.BF16: every gradient handed to step() is in the library's 16-bit type (16-bit buckets)
        self._factored = None         # [(param index, dy [T, N] fp32, a [T, K])]: gradients formed on the fly (set_factored)
        self._extra_sq = torch.zeros(8, dtype=torch.float32, device=self.device)
        self._ext_grads = None        # index -> tensor: gradients read from there instead of p.grad (torch refuses a .grad whose
                                      # dtype differs from the parameter's, so 16-bit bucket views cannot be p.grad)
        # tables
        wd_of = {id(p): float(g["weight_decay"]) for g in self.param_groups for p in g["params"]}
        tt = (L.OptTensor * len(params))()
        chunk_tensor, chunk_off = [], []
        self._shadow_refs = []
        for i, p in enumerate(params):
            sh = shadows.get(id(p))
            if sh is not None:
                assert sh.dtype == L.half_dtype() and sh.numel() == p.numel() and sh.is_contiguous() and sh.device == p.device
                self._shadow_refs.append(sh)
                self.__dict__.setdefault("_shadow_by_index", {})[i] = sh
            tt[i].p, tt[i].m, tt[i].v = p.data_ptr(), self._m[i].data_ptr(), self._v[i].data_ptr()
            tt[i].w16 = sh.data_ptr() if sh is not None else None
            tt[i].n, tt[i].weight_decay = p.numel(), wd_of[id(p)]
            for c0 in range(0, p.numel(), CHUNK):
                chunk_tensor.append(i)
                chunk_off.append(c0)
        self._param_ptrs = [p.data_ptr() for p in params]
        self.nchunks = len(chunk_tensor)
        self._tensors = torch.frombuffer(bytearray(bytes(tt)), dtype=torch.uint8).to(self.device)
        self._chunk_tensor = torch.tensor(chunk_tensor, dtype=torch.int32, device=self.device)
        self._chunk_off = torch.tensor(chunk_off, dtype=torch.int64, device=self.device)
        self._partial = torch.empty(self.nchunks, dtype=torch.float32, device=self.device)
        # gradient addresses change from step to step in eager mode: they travel through a small ring of pinned host
        # buffers (the host may run a full step ahead of the GPU, so a slot is reused only after its copy has executed);
        # a captured graph gets a pinned buffer of its own that is never written again.
        self._ring = [torch.zeros(len(params), dtype=torch.int64).pin_memory() for _ in range(4)]
        self._ring_ev = [None] * len(self._ring)
        self._ring_pos = 0
        self._capture_pool = [torch.zeros(len(params), dtype=torch.int64).pin_memory() for _ in range(4)]
        self._captured_hosts = []
        self._grads_dev = torch.zeros(len(params), dtype=torch.int64, device=self.device)
        self.n_total = n_total

    def refill_capture_pool(self):
        """Top the pinned capture buffers up again (outside a capture): one process may capture step() any number of times."""
        while len(self._capture_pool) < 4:
            self._capture_pool.append(torch.zeros(len(self.params), dtype=torch.int64).pin_memory())

    # ------------------------------------------------------------------ torch.optim-like surface
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @property
    def grad_norm(self) -> torch.Tensor:
        """Total L2 gradient norm of the last step (device scalar; what clip_grad_norm_ returns)."""
        return self.state_t[1]

    # ------------------------------------------------------------------ factored gradients (fusion convs)
    FACTORED_MAX_T = 64
    FACTORED_MAX_T_16 = 256     # with 16-bit operands (MFMA form of the update + Gram norm through the library GEMM)

    def set_factored(self, items=None):
        """items: [(param, dy [T, N] fp32, a [T, K] fp32 / 16-bit)] with param.view(N, K) -- the weight gradient dY^T A of these
        parameters is never materialised: step() adds its squared norm to the clip norm from T x T Gram matrices
        (csts_factored_sqnorm) and forms g on the fly inside their AdamW update (csts_adamw_factored).  Their p.grad is ignored.
        None / [] switches back.  Tensors must stay alive and in place while step() may run (a captured graph: for its lifetime)."""
        if not items:
            self._factored = None
            return
        idx = {id(p): i for i, p in enumerate(self.params)}
        fac = []
        for p, dy, a in items:
            i = idx[id(p)]
            N = p.shape[0]
            K = p.numel() // N
            T = dy.shape[0]
            if not (dy.dtype == torch.float32 and dy.is_contiguous() and a.is_contiguous() and tuple(dy.shape) == (T, N) and tuple(a.shape) == (T, K)
                    and self.factored_ok(p, T, a.dtype != torch.float32)):
                raise L.CstsError("factored gradient: dy [T, N] fp32 and a [T, K] contiguous, T <= 64 (256 with 16-bit a), K % 256 == 0, N % 16 == 0")
            fac.append((i, dy, a))
        if len(fac) > 8:
            raise L.CstsError("at most 8 factored parameters")
        self._factored = fac
        self.last_factored_params = [self.params[i] for i, _, _ in fac]     # introspection: which weights the last step updated from factors

    @staticmethod
    def factored_ok(p, T, a16: bool = True):
        """Can a gradient dY[T, N]^T A[T, K] of p stay factored?  T <= 64 always; up to FACTORED_MAX_T_16 = 256 rows when A is in the
        library's 16-bit type (the MFMA form of the update loops over T; the Gram norm then goes through the library GEMM): the
        data-parallel chain's W * B * T' gathered rows at eight ranks."""
        N = p.shape[0]
        K = p.numel() // N
        lim = FusedAdamW.FACTORED_MAX_T_16 if (a16 and K % 64 == 0 and N * K * 4 < 2 ** 31) else FusedAdamW.FACTORED_MAX_T
        return T <= lim and K % 256 == 0 and N % 16 == 0

    def _factored_items(self):
        arr = (L.OptFactored * len(self._factored))()
        wd_of = {id(p): float(g["weight_decay"]) for g in self.param_groups for p in g["params"]}
        sh = {i: s for i, s in getattr(self, "_shadow_by_index", {}).items()}
        for j, (i, dy, a) in enumerate(self._factored):
            p = self.params[i]
            arr[j].p, arr[j].m, arr[j].v = p.data_ptr(), self._m[i].data_ptr(), self._v[i].data_ptr()
            arr[j].w16 = sh[i].data_ptr() if i in sh else None
            arr[j].dy, arr[j].a = dy.data_ptr(), a.data_ptr()
            arr[j].a_dt = L.F32 if a.dtype == torch.float32 else L.BF16
            arr[j].N, arr[j].K, arr[j].T = p.shape[0], p.numel() // p.shape[0], dy.shape[0]
            arr[j].weight_decay = wd_of[id(p)]
        return arr

    def set_external_grads(self, grads_by_param=None, grad_dt=None):
        """grads_by_param: {id(param): tensor} -- step() reads these instead of p.grad (data-parallel buckets in 16 bits);
        None restores p.grad.  grad_dt: L.F32 | L.BF16 (the library's 16-bit type) of EVERY gradient."""
        if grads_by_param is None:
            self._ext_grads, self.grad_dt = None, L.F32
            return
        self._ext_grads = {i: grads_by_param[id(p)] for i, p in enumerate(self.params) if id(p) in grads_by_param}
        if grad_dt is not None:
            self.grad_dt = grad_dt

    @torch.no_grad()
    def step(self):
        ptrs = []
        fac_idx = {i for i, _, _ in self._factored} if self._factored else ()
        for i, p in enumerate(self.params):
            g = p.grad if self._ext_grads is None else self._ext_grads.get(i)
            if g is None or i in fac_idx:          # factored parameters: updated by csts_adamw_factored below, skipped here
                ptrs.append(0)
                continue
            if g.dtype != (torch.float32 if self.grad_dt == L.F32 else L.half_dtype()) or not g.is_contiguous():
                raise L.CstsError("FusedAdamW needs contiguous gradients of its grad_dt (fp32, or all 16-bit)")
            if p.data_ptr() != self._param_ptrs[i]:
                raise L.CstsError("a parameter was re-allocated after the optimizer was built (rebuild the optimizer)")
            ptrs.append(g.data_ptr())
        if torch.cuda.is_current_stream_capturing():
            if not self._capture_pool:
                raise L.CstsError("FusedAdamW: out of pinned capture buffers (call refill_capture_pool() outside the capture)")
            host = self._capture_pool.pop()      # pre-allocated: no host allocation while a capture is open
            host.copy_(torch.tensor(ptrs, dtype=torch.int64))
            self._captured_hosts.append(host)
            self._grads_dev.copy_(host, non_blocking=True)
            grads_dev = self._grads_dev
        else:
            grads_dev = self._grads_dev
            k = self._ring_pos
            self._ring_pos = (k + 1) % len(self._ring)
            if self._ring_ev[k] is not None:
                self._ring_ev[k].synchronize()
            self._ring[k].copy_(torch.tensor(ptrs, dtype=torch.int64))
            self._grads_dev.copy_(self._ring[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._ring_ev[k] = ev
        a = L.OptArgs()
        a.chunk_tensor, a.chunk_off = self._chunk_tensor.data_ptr(), self._chunk_off.data_ptr()
        a.nchunks, a.chunk_elems = self.nchunks, CHUNK
        a.tensors, a.grads, a.ntensors = self._tensors.data_ptr(), grads_dev.data_ptr(), len(self.params)
        a.partial, a.state, a.lr = self._partial.data_ptr(), self.state_t.data_ptr(), self._lr.data_ptr()
        a.beta1, a.beta2, a.eps, a.max_grad_norm = self.betas[0], self.betas[1], self.eps, self.max_grad_norm
        a.grad_dt = self.grad_dt
        if self.scaler_t is not None:
            a.scaler = self.scaler_t.data_ptr()
            a.growth, a.backoff, a.growth_interval = self.scaler_cfg
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        items = None
        if self._factored:
            items = self._factored_items()
            n = len(self._factored)
            if max(dy.shape[0] for _, dy, _ in self._factored) <= self.FACTORED_MAX_T:
                nb = int(lib.csts_factored_sqnorm_workspace(items, n))
                ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=self.device)
                L.check(lib.csts_factored_sqnorm(items, n, self._extra_sq.data_ptr(), ws.data_ptr(), ws.numel(), st), "csts_factored_sqnorm")
            else:
                self._factored_sqnorm_gemm(st)
            a.extra_sq, a.n_extra_sq = self._extra_sq.data_ptr(), n
        L.check(lib.csts_adamw_step(C.byref(a), st), "csts_adamw_step")
        if items is not None:
            L.check(lib.csts_adamw_factored(items, len(self._factored), self.state_t.data_ptr(), self._lr.data_ptr(), self.betas[0],
                                            self.betas[1], self.eps, st), "csts_adamw_factored")

    def _factored_sqnorm_gemm(self, st):
        """||dY^T A||_F^2 = sum_{t,t'} (dY dY^T)[t,t'] (A A^T)[t,t'] for more than 64 token rows (the data-parallel chain at W > 2 ranks:
        T = W * B * T'): both T x T Gram matrices on the matrix cores through the library's own NT GEMM (A A^T: K = 49152 deep, split-K;
        dY rounded to the 16-bit type first -- the gradient the MFMA update forms is that of the rounded dY), then their inner product
        with csts_rowdot2 + csts_reduce_rows.  Fixed summation orders: reproducible."""
        from . import ops
        lib = L.load()
        half = L.half_dtype()
        for j, (i, dy, a) in enumerate(self._factored):
            T, N = dy.shape
            K = a.shape[1]
            g2 = torch.empty(T, T, dtype=torch.float32, device=self.device)
            ops.gemm(L.GEMM_NT, a, 0, K, a, 0, K, g2, T, T, T, K, compute=L.BF16, split_k=max(1, min(128, K // 512)))
            dy16 = torch.empty(T, N, dtype=half, device=self.device)
            L.check(lib.csts_axpby(dy.data_ptr(), L.F32, None, L.F32, dy16.data_ptr(), L.BF16, T * N, 1.0, 0.0, st), "csts_axpby")
            g1 = torch.empty(T, T, dtype=torch.float32, device=self.device)
            ops.gemm(L.GEMM_NT, dy16, 0, N, dy16, 0, N, g1, T, T, T, N, compute=L.BF16)
            r = torch.empty(T, dtype=torch.float32, device=self.device)
            L.check(lib.csts_rowdot2(g1.data_ptr(), L.F32, g2.data_ptr(), L.F32, r.data_ptr(), T, T, st), "csts_rowdot2")
            L.check(lib.csts_reduce_rows(r.data_ptr(), self._extra_sq.data_ptr() + 4 * j, T, 1, 1.0, st), "csts_reduce_rows")

    @property
    def loss_scale(self):
        """Device scalar the loss is multiplied by before backward (None without loss scaling)."""
        return self.scaler_t[0] if self.scaler_t is not None else None

    def scaler_state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict() layout (the reference stores it as "scaler_state", checkpoint.py:133-134)."""
        if self.scaler_t is None:
            return {}
        sc, tr = self.scaler_t.tolist()
        g, b, gi = self.scaler_cfg
        return {"scale": sc, "growth_factor": g, "backoff_factor": b, "growth_interval": gi, "_growth_tracker": int(tr)}

    def load_scaler_state_dict(self, sd):
        if self.scaler_t is None or not sd:
            return
        self.scaler_t.copy_(torch.tensor([float(sd["scale"]), float(sd.get("_growth_tracker", 0))]))
        self.scaler_cfg = (float(sd.get("growth_factor", self.scaler_cfg[0])), float(sd.get("backoff_factor", self.scaler_cfg[1])),
                           int(sd.get("growth_interval", self.scaler_cfg[2])))

    def step_count(self) -> int:
        """AdamW steps taken so far (host sync: logging / checkpoint bookkeeping only)."""
        return int(float(self.state_t[0]))

    def reset_state(self):
        """Forget the moments and the step count (a fresh optimizer over the same parameters)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.state_t.zero_()

    def state_dict(self):
        """torch.optim.AdamW's layout (what the reference stores as "optimizer_state", checkpoint.py:131): per-parameter
        step / exp_avg / exp_avg_sq keyed by the parameter's index over the param groups, plus the groups' hyper-parameters."""
        step = self.state_t[0:1].detach().clone().reshape(())
        state = {i: {"step": step.clone(), "exp_avg": self._m[i].detach().clone().view_as(p),
                     "exp_avg_sq": self._v[i].detach().clone().view_as(p)} for i, p in enumerate(self.params)}
        groups, off = [], 0
        for g in self.param_groups:
            n = len(g["params"])
            groups.append({"lr": float(self._lr), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": g["weight_decay"],
                           "amsgrad": False, "params": list(range(off, off + n))})
            off += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        state = sd["state"]
        if len(state) not in (0, len(self.params)):
            raise ValueError("optimizer state has a different number of parameters")
        steps = set()
        for i, p in enumerate(self.params):       # refuse a state whose tensors do not fit the parameters at the same positions
            st = state.get(i, state.get(str(i)))
            if st is not None and (st["exp_avg"].numel() != p.numel() or st["exp_avg_sq"].numel() != p.numel()):
                raise ValueError(f"optimizer state {i}: {tuple(st['exp_avg'].shape)} does not fit parameter {tuple(p.shape)} "
                                 "(parameter groups / order differ from the run that wrote the checkpoint)")
        for i, p in enumerate(self.params):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            self._m[i].copy_(st["exp_avg"].reshape(-1))
            self._v[i].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(float(st["step"]))
        if steps:
            # one step counter for the whole set (every parameter of this path receives a gradient on every iteration)
            self.state_t[0:1].fill_(max(steps))
        if sd.get("param_groups"):
            self._lr.fill_(float(sd["param_groups"][0]["lr"]))
