"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, host logic (config, registry,
state_dict surface, LR schedule, optimizer groups, CLI parsing), loud failure without a GPU, and the N>1 data-parallel
logic over a 2-process gloo group."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
YAML = os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml")


def test_library_exports_every_declared_symbol():
    from csts_amd import lib
    handle = lib.load()                      # raises loudly if the .so is missing / stale
    assert handle.csts_abi_version() == lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "csts_hip.h")).read()
    declared = set(re.findall(r"\b(csts_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(lib.SYMBOLS), (declared ^ set(lib.SYMBOLS))
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.csts_last_error() is not None
    assert handle.csts_half_kind() == 0 and lib.half_dtype() == torch.bfloat16
    m = re.search(r"#define CSTS_ABI_VERSION (\d+)", hdr)
    assert m and int(m.group(1)) == lib.ABI_VERSION           # the binding mirrors THIS header's struct layouts
    with pytest.raises(lib.CstsError):                        # one 16-bit kernel library per process
        lib.set_half("fp16")


def test_fp16_library_exports_every_declared_symbol_in_a_fresh_process():
    """libcsts_hip_f16.so (the same sources with IEEE half as the 16-bit type, CSTS_AMD.COMPUTE fp16) loads, exports every
    declared symbol and reports its kind; selected per process (CSTS_HALF / lib.set_half)."""
    code = ("import torch; from csts_amd import lib; lib.set_half('fp16'); h = lib.load(); "
            "assert h.csts_half_kind() == 1 and h.csts_abi_version() == lib.ABI_VERSION and lib.half_dtype() == torch.float16; "
            "assert all(hasattr(h, n) for n in lib.SYMBOLS); print('ok', len(lib.SYMBOLS))")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("ok"), p.stderr[-2000:]


def test_bad_call_is_rejected_with_message_and_no_gpu_needed():
    """Argument validation happens on the host before any launch: a null problem returns -1 + a message."""
    from csts_amd import lib
    h = lib.load()
    a = lib.GemmArgs()
    rc = h.csts_gemm(a, None)
    assert rc != 0 and b"csts_gemm" in h.csts_last_error()
    assert h.csts_layernorm_bwd_workspace(1024, 96) > 0
    assert h.csts_colsum_workspace(1, 4096, 96) > 0


def test_registry_and_build_boundary():
    import slowfast.models as sm
    from csts_amd.config import load_yaml
    assert sm.MODEL_REGISTRY.get("CSTS").__name__ == "CSTS"
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "MODEL.LOSS_FUNC", "kldiv+egonce"])
    m = sm.build_model(cfg)
    ref = json.load(open(os.path.join(GOLDEN, "manifest_T8.json")))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["entries"]
    assert sum(p.numel() for p in m.parameters()) == ref["num_params"] == 188182401
    assert m.no_weight_decay() == {}
    # reference mutates cfg.MVIT.POOL_KV_STRIDE in __init__ (custom_multimodal_builder.py:136-142); so do we
    geo = json.load(open(os.path.join(GOLDEN, "geometry_T8.json")))
    assert cfg.MVIT.POOL_KV_STRIDE == geo["pool_kv_stride"]
    # init contract: LN weight 1 / bias 0, Linear bias 0, trunc-normal(0.02) Linear weights
    assert float(m.blocks[0].norm1.weight.min()) == 1.0 and float(m.blocks[3].mlp.fc1.bias.abs().max()) == 0.0
    assert 0.015 < float(m.blocks[3].mlp.fc1.weight.std()) < 0.022
    with pytest.raises(KeyError):
        sm.MODEL_REGISTRY.get("SlowFast")


def test_kldiv_only_variant_and_T32_manifest():
    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    m = build_model(load_yaml(YAML, ["NUM_GPUS", 0]))       # YAML default LOSS_FUNC kldiv: no vision_proj/audio_proj
    ref = json.load(open(os.path.join(GOLDEN, "manifest_T8_kldiv.json")))
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["entries"]
    m32 = build_model(load_yaml(YAML, ["NUM_GPUS", 0, "DATA.NUM_FRAMES", 32, "MODEL.LOSS_FUNC", "kldiv+egonce"]))
    ref32 = json.load(open(os.path.join(GOLDEN, "manifest_T32.json")))
    assert [[k, list(v.shape)] for k, v in m32.state_dict().items()] == ref32["entries"]


def test_forward_fails_loudly_without_gpu():
    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    from csts_amd.lib import CstsError
    m = build_model(load_yaml(YAML, ["NUM_GPUS", 0]))
    with pytest.raises(CstsError):
        m([torch.zeros(1, 3, 8, 256, 256)], torch.zeros(1, 1, 8, 256, 256))
    from csts_amd import ops
    with pytest.raises(CstsError):
        ops.layer_norm(torch.zeros(4, 96), torch.ones(96), torch.zeros(96), 1e-6, 0)


def test_unsupported_configs_raise_like_reference():
    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    with pytest.raises(NotImplementedError):
        build_model(load_yaml(YAML, ["NUM_GPUS", 0, "MVIT.NORM", "batchnorm"]))
    with pytest.raises(AssertionError):
        build_model(load_yaml(YAML, ["NUM_GPUS", 0, "DATA.TEST_CROP_SIZE", 224]))   # custom_multimodal_builder.py:28


def test_lr_schedule_matches_reference_samples():
    from csts_amd.config import load_yaml
    from csts_amd.train import get_lr_at_epoch
    cfg = load_yaml(YAML, ["NUM_GPUS", 0])
    g = np.load(os.path.join(GOLDEN, "lr_schedule.npz"))
    for e, lr in zip(g["epochs"], g["lr"]):
        assert abs(get_lr_at_epoch(cfg, float(e)) - float(lr)) < 1e-12


def test_optimizer_param_groups():
    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    from csts_amd.train import construct_optimizer
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "MODEL.LOSS_FUNC", "kldiv+egonce"])
    m = build_model(cfg)
    opt = construct_optimizer(m, cfg)
    wd = {g["weight_decay"]: sum(p.numel() for p in g["params"]) for g in opt.param_groups}
    assert set(wd) == {0.05, 0.0}
    n1d = sum(p.numel() for n, p in m.named_parameters() if p.dim() == 1 or n.endswith(".bias"))
    assert wd[0.0] == n1d and wd[0.05] + wd[0.0] == 188182401
    assert opt.defaults["eps"] == 1e-8
    # pos-embeds (3-D) DO get weight decay with ZERO_DECAY_POS_CLS False
    assert any(p is m.pos_embed_spatial for g in opt.param_groups if g["weight_decay"] > 0 for p in g["params"])


def test_cli_parsing_and_yaml_tuple_strings():
    from csts_amd.cli import parse_args, load_config
    args = parse_args(["--cfg", YAML, "--init_method", "tcp://127.0.0.1:9997", "NUM_GPUS", "8", "TRAIN.BATCH_SIZE", "32",
                       "MODEL.LOSS_FUNC", "kldiv+egonce", "OUTPUT_DIR", "/tmp/csts_test_out"])
    cfg = load_config(args)
    assert cfg.NUM_GPUS == 8 and cfg.TRAIN.BATCH_SIZE == 32 and cfg.MODEL.LOSS_FUNC == "kldiv+egonce"
    assert cfg.MVIT.PATCH_KERNEL == [3, 7, 7] and cfg.SOLVER.COSINE_END_LR == 1e-6
    for y in ("configs/Aria/CSTS_Aria_Gaze_Forecast.yaml", "configs/Aria/CSTS_Aria_Gaze_Estimation.yaml",
              "configs/Ego4D/CSTS_Ego4D_Gaze_Estimation.yaml"):
        c = load_config(parse_args(["--cfg", os.path.join(ROOT, y), "OUTPUT_DIR", "/tmp/csts_test_out"]))
        assert c.MODEL.MODEL_NAME == "CSTS" and c.DATA.NUM_FRAMES == 8 and c.DATA.TRAIN_CROP_SIZE == 256


# ----------------------------------------------------------------------------------------------- 2-rank gloo
def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from csts_amd import distributed as du
        from oracle import csts_oracle as O
        torch.manual_seed(0)
        # (1) EgoNCE embedding gather: W-rank gradient == single-process gradient on the concatenated batch
        full_v = torch.randn(4, 16)
        full_a = torch.randn(4, 16)
        v = full_v[2 * rank:2 * rank + 2].clone().requires_grad_(True)
        a = full_a[2 * rank:2 * rank + 2].clone().requires_grad_(True)
        gv, ga = du.all_gather_with_grad([v, a])
        assert torch.equal(gv.detach(), full_v)
        loss = O.egonce(O.sim_matrix(gv, ga))
        loss.backward()
        fv, fa = full_v.clone().requires_grad_(True), full_a.clone().requires_grad_(True)
        O.egonce(O.sim_matrix(fv, fa)).backward()
        # data-parallel convention: per-rank grads are averaged over ranks afterwards
        ok1 = torch.allclose(v.grad / world, fv.grad[2 * rank:2 * rank + 2], atol=1e-6)
        # (2) bucketed gradient all-reduce == mean of per-rank grads, all buckets flushed, ready-order re-bucketing
        torch.manual_seed(1)
        net = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                  torch.nn.Linear(32, 4))
        wrapped = du.GradAllReduce(net, bucket_mb=0)      # 0 MB -> one bucket per parameter: exercises ordering
        ok2 = True
        for it in range(2):
            x = torch.randn(3, 8, generator=torch.Generator().manual_seed(100 + rank + 10 * it))
            for p in net.parameters():
                p.grad = None
            wrapped(x).pow(2).sum().backward()
            local = [p.grad.clone() for p in net.parameters()]
            wrapped.finish()
            for p, l in zip(net.parameters(), local):
                buf = [torch.empty_like(l) for _ in range(world)]
                dist.all_gather(buf, l)
                ok2 &= torch.allclose(p.grad, sum(buf) / world, atol=1e-6)
        # (3) fused scalar all-reduce and metric gather
        r = du.all_reduce([torch.tensor(float(rank)), torch.tensor(2.0 * rank)])
        ok3 = abs(float(r[0]) - 0.5) < 1e-6 and abs(float(r[1]) - 1.0) < 1e-6
        g = du.all_gather([torch.full((2, 3), float(rank))])[0]
        ok3 &= g.shape == (4, 3) and float(g[0, 0]) == 0.0 and float(g[3, 0]) == 1.0
        q.put((rank, bool(ok1), bool(ok2), bool(ok3)))
    finally:
        dist.destroy_process_group()


class _TinyBlock(torch.nn.Module):
    """CSTS-shaped parameter names (norm1 / attn.qkv / attn.pool_k / attn.norm_k / norm2 / mlp.fc1 / mlp.fc2) over plain
    torch CPU ops: what GradAllReduce sees of a block."""

    def __init__(self, dim):
        super().__init__()
        self.norm1 = torch.nn.LayerNorm(dim)
        self.attn = torch.nn.Module()
        self.attn.qkv = torch.nn.Linear(dim, dim)
        self.attn.pool_k = torch.nn.Conv1d(dim, dim, 3, padding=1, groups=dim, bias=False)
        self.attn.norm_k = torch.nn.LayerNorm(dim)
        self.norm2 = torch.nn.LayerNorm(dim)
        self.mlp = torch.nn.Module()
        self.mlp.fc1 = torch.nn.Linear(dim, 2 * dim)
        self.mlp.fc2 = torch.nn.Linear(2 * dim, dim)

    def forward(self, x):
        h = self.attn.qkv(self.norm1(x))
        h = self.attn.norm_k(self.attn.pool_k(h.transpose(1, 2)).transpose(1, 2))
        x = x + h
        return x + self.mlp.fc2(torch.relu(self.mlp.fc1(self.norm2(x))))


class _TinyCSTS(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = torch.nn.ModuleList([_TinyBlock(8) for _ in range(2)])
        self.blocks_audio = torch.nn.ModuleList([_TinyBlock(8)])
        self.vision_pool = torch.nn.Linear(8, 8)
        self.classifier = torch.nn.Linear(8, 1)

    def forward(self, x, y):
        for b in self.blocks:
            x = b(x)
        for b in self.blocks_audio:
            y = b(y)
        return self.classifier(self.vision_pool(x) + y.mean(dim=1, keepdim=True))


class _StubStream:
    """Stands in for a HIP stream on the CPU: records who waited for whom."""
    def __init__(self, sid, log):
        self.cuda_stream, self._log = sid, log

    def wait_stream(self, other):
        self._log.append((self.cuda_stream, other.cuda_stream))


def _dist_worker_whole(rank, world, port, q):
    """The WHOLE GradAllReduce on a CSTS-shaped module, 2 ranks over gloo: hook-driven buckets in gradient-ready order, the
    late bucket (LayerNorm / pooling-stencil parameters, reduced by finish()), re-bucketing after the first iteration, and
    the cross-stream wait logic (audio-trunk gradients are produced on another stream) behind stub streams."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from csts_amd import distributed as du
        from csts_amd import ops
        torch.manual_seed(3)
        net = _TinyCSTS()
        waits = []
        main, side = _StubStream(1, waits), _StubStream(2, waits)
        audio = list(net.blocks_audio.parameters())
        audio_ids = {id(p) for p in audio}

        class Wrapped(du.GradAllReduce):
            def _cur_stream(self, grad):       # audio-trunk gradients are "produced on the side stream"
                return side if any(grad is p_.grad for p_ in audio) else main

        w = Wrapped(net, bucket_mb=0)
        w.bucket_bytes = 700                   # a few parameters per bucket: video and audio gradients share buckets
        w._assign(list(reversed(w._params)))
        want_late = {n for n, _ in net.named_parameters() if any(t in n for t in (".norm1.", ".norm2.", ".norm_k.", ".pool_k."))}
        ok_late = {n for n, p in net.named_parameters() if any(p is l for l in w._late)} == want_late and len(want_late) == 21
        ok = True
        for it in range(3):
            gen = torch.Generator().manual_seed(50 + rank + 7 * it)
            x, y = torch.randn(2, 5, 8, generator=gen), torch.randn(2, 5, 8, generator=gen)
            for p in net.parameters():
                p.grad = None
            w(x, y).pow(2).sum().backward()
            local = {n: p.grad.clone() for n, p in net.named_parameters()}
            w.finish()
            for n, p in net.named_parameters():
                buf = [torch.empty_like(local[n]) for _ in range(world)]
                dist.all_gather(buf, local[n])
                ok = ok and bool(torch.allclose(p.grad, sum(buf) / world, atol=1e-6))
        ok_order = w._observed and len(w._ready_order) == len(w._params) and not w._pending
        # a bucket that holds both video and audio gradients is launched from one stream and must wait for the other
        mixed = [bi for bi, b_ in enumerate(w._buckets) if 0 < sum(id(p) in audio_ids for p in b_) < len(b_)]
        ok_wait = len(mixed) > 0 and len(waits) > 0 and all(a_ != b_ for a_, b_ in waits)
        ok_mode = ops.GROUP_WGRADS == "never"            # hook-driven buckets read gradients mid-backward: no grouped tail
        q.put((rank, bool(ok), bool(ok_late), bool(ok_order), bool(ok_wait), bool(ok_mode), len(mixed), len(waits)))
    except Exception as e:       # surface the failure instead of a queue timeout
        q.put((rank, False, False, False, False, False, repr(e), 0))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_whole_grad_allreduce_on_csts_shaped_module():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dist_worker_whole, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1], "averaged gradients != mean of the per-rank gradients"
        assert r[2], "late bucket is not exactly the LayerNorm / pooling-stencil parameters"
        assert r[3], "ready-order re-bucketing / pending buckets"
        assert r[4], ("cross-stream waits", r[6], r[7])
        assert r[5], "GradAllReduce must switch the grouped weight-gradient tail off"


class _ToySegModel(torch.nn.Module):
    """A model with the CSTS segment interface (forward(x, y, return_embed, boundary) / head_parameters /
    early_trunk_parameters) over plain torch CPU ops: trunk = a two-stage video encoder whose first-stage feature is also a
    decoder skip + an audio encoder, head = fusion + 'decoder' + embedding projections."""

    def __init__(self):
        super().__init__()
        self.enc_v0 = torch.nn.Linear(6, 16)
        self.enc_v1 = torch.nn.Sequential(torch.nn.GELU(), torch.nn.Linear(16, 16))
        self.enc_a = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 16))
        self.fuse = torch.nn.Linear(32, 16)
        self.dec = torch.nn.Linear(16, 5)
        self.vision_proj = torch.nn.Linear(16, 4)
        self.audio_proj = torch.nn.Linear(16, 4)

    def head_parameters(self):
        return [p for m in (self.fuse, self.dec, self.vision_proj, self.audio_proj) for p in m.parameters()]

    def early_trunk_parameters(self, k):
        return list(self.enc_v0.parameters())

    def forward(self, x, y, return_embed=False, boundary=None):
        e = self.enc_v0(x[0])
        skip = e * 1.0                                     # the early feature the decoder re-uses
        if getattr(boundary, "trunk_cut", 0):
            e = boundary.inner([e])[0]
        feats = [self.enc_v1(e), self.enc_a(y), skip]
        if boundary is not None:
            feats = boundary(feats)
        v, a, skip = feats
        logits = self.dec(torch.tanh(self.fuse(torch.cat([v, a], dim=-1))) + 0.5 * skip)
        return [logits, self.vision_proj(v.mean(dim=1)), self.audio_proj(a.mean(dim=1))]


def _toy_loss(gathered_fn):
    def loss_fn(leaves, static):
        logits, v, a = leaves
        v, a = gathered_fn([v, a])
        kld = (logits - static["labels_hm"]).pow(2).mean()
        sim = torch.nn.functional.normalize(v, dim=1) @ torch.nn.functional.normalize(a, dim=1).t()
        nce = -(torch.log_softmax(sim / 0.05, dim=1).diag().mean() + torch.log_softmax(sim / 0.05, dim=0).diag().mean())
        return kld + 0.05 * nce, kld, nce
    return loss_fn


def _dist_worker_segmented(rank, world, port, q, trunk_cut=0, bucket_dtype="fp32"):
    """csts_amd.train.SegmentedTrainStep (the data-parallel step: forward | eager losses + embedding all-gather | backward
    head | all-reduce(head bucket) | backward trunk | all-reduce(trunk bucket) | optimizer) run WITHOUT graphs on 2 gloo
    ranks: the gradients the optimizer sees must equal those of ONE process on the concatenated batch, p.grad must be
    views of the two flat buckets, and the replicas must stay identical after the step."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from csts_amd import distributed as du
        from csts_amd import train as T
        from csts_amd.config import load_yaml
        cfg = load_yaml(YAML, ["NUM_GPUS", 0, "MODEL.LOSS_FUNC", "kldiv+egonce", "SOLVER.CLIP_GRAD_L2NORM", None,
                               "CSTS_AMD.GRAD_BUCKET_DTYPE", bucket_dtype])
        torch.manual_seed(7)
        net = _ToySegModel()
        g = torch.Generator().manual_seed(123)
        nb = 2 * world
        full = {"video": torch.randn(nb, 3, 6, generator=g), "audio": torch.randn(nb, 3, 6, generator=g),
                "labels_hm": torch.randn(nb, 3, 5, generator=g)}
        mine = {k: v[2 * rank:2 * rank + 2] for k, v in full.items()}
        # single-process reference on the concatenated batch (mean over the global batch = mean of the per-rank means)
        import copy
        ref = copy.deepcopy(net)
        out = ref([full["video"]], full["audio"], return_embed=True)
        lr_ = [(out[0][2 * r:2 * r + 2] - full["labels_hm"][2 * r:2 * r + 2]).pow(2).mean() for r in range(world)]
        loss_ref, _, nce_ref = _toy_loss(lambda ts: ts)([out[0], out[1], out[2]], {"labels_hm": full["labels_hm"]})
        (sum(lr_) / world + 0.05 * nce_ref).backward()
        wrapped = du.GradAllReduce(net, bucket_mb=1)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        seg = T.SegmentedTrainStep(cfg, wrapped, opt, mine, use_graphs=False, loss_fn=_toy_loss(du.all_gather_with_grad),
                                   trunk_cut=trunk_cut)
        assert len(seg.flat) == (3 if trunk_cut else 2) and sum(len(b) for b in seg.buckets) == len(list(net.parameters()))
        before = [p.detach().clone() for p in net.parameters()]
        loss, kld, nce = seg.step_eager(mine, lr=0.1)
        ok_hooks = not wrapped.hooks_enabled
        flat_rng = [(f.data_ptr(), f.data_ptr() + 4 * f.numel()) for f, _ in seg.flat]
        ok_view = all(any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_rng) for p in net.parameters())
        if bucket_dtype == "fp32":
            close = lambda x, y: torch.allclose(x, y, atol=1e-6)
        else:       # buckets travelled in bf16: equal to bf16 rounding (two roundings: the cast and the sum over ranks)
            assert seg.bucket16 and seg.flat16[0][0].dtype == torch.bfloat16
            close = lambda x, y: bool(((x - y).abs() <= 2.0 ** -6 * y.abs() + 2.0 ** -8 * y.abs().max() + 1e-7).all())
        ok_grad = all(close(p.grad, r.grad) for p, r in zip(net.parameters(), ref.parameters()))
        ok_nce = abs(float(nce) - float(nce_ref)) < 1e-5          # every rank sees the global similarity matrix
        ok_step = all(close((b - p) / 0.1, r.grad) for p, b, r in zip(net.parameters(), before, ref.parameters()))
        flatw = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.empty_like(flatw) for _ in range(world)]
        dist.all_gather(both, flatw)
        ok_same = all(torch.equal(both[0], b_) for b_ in both[1:])
        q.put((rank, ok_hooks, ok_view, ok_grad, ok_nce, ok_step, ok_same))
    except Exception as e:
        import traceback
        q.put((rank, False, False, False, False, False, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("trunk_cut", [0, 1])
def test_two_rank_gloo_segmented_data_parallel_step(trunk_cut):
    """trunk_cut = 1: the three-bucket chain (head | late trunks | early video trunk behind a second autograd cut)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + 2500 * trunk_cut
    procs = [ctx.Process(target=_dist_worker_segmented, args=(r, 2, port, q, trunk_cut)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1], ("hook-driven buckets must be off", r)
        assert r[2], "p.grad must be views of the flat buckets"
        assert r[3], ("averaged gradients != single-process gradients on the concatenated batch", r[6])
        assert r[4], "EgoNCE over the gathered embeddings"
        assert r[5] and r[6] is True, ("optimizer step / replicas diverged", r)


@pytest.mark.parametrize("bucket_dtype", ["fp32", "bf16"])
def test_four_rank_gloo_segmented_step_and_16bit_buckets(bucket_dtype):
    """W = 4: every rank takes ITS OWN rows of the gathered EgoNCE embeddings' gradient, the averaged
    bucket gradients equal ONE process on the concatenated batch of 8 -- exactly with fp32 buckets, to bf16 rounding with
    CSTS_AMD.GRAD_BUCKET_DTYPE bf16 (the buckets travel in 16 bits, half the xGMI bytes) -- and the replicas stay identical."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + (os.getpid() % 2000) + (700 if bucket_dtype == "bf16" else 0)
    procs = [ctx.Process(target=_dist_worker_segmented, args=(r, 4, port, q, 1, bucket_dtype)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1, 2, 3]
    for r in res:
        assert r[1] and r[2], r
        assert r[3], ("averaged gradients != single-process gradients on the concatenated batch", r[6])
        assert r[4], "EgoNCE over the gathered embeddings"
        assert r[5] and r[6] is True, ("optimizer step / replicas diverged", r)


def test_two_rank_gloo_data_parallel_logic():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1], "all_gather_with_grad backward != single-process gradient"
        assert r[2], "bucketed gradient all-reduce != mean of per-rank gradients"
        assert r[3], "scalar all-reduce / gather"


def test_checkpoint_wire_format_reads_reference_written_pyth(tmp_path):
    """csts_amd.checkpoint against a .pyth written by the REFERENCE's save_checkpoint (fixture from oracle/gen_golden.py):
    name + shape matching, bilinear pos-embed resize, epoch / optimizer-state handling == the reference's load_checkpoint
    (slowfast/utils/checkpoint.py:185-354), and our own save_checkpoint round-trips through the same format."""
    import numpy as np
    from csts_amd import checkpoint as ck
    from csts_amd.config import load_yaml

    class Tiny(torch.nn.Module):
        def __init__(self, T, nout):
            super().__init__()
            self.pos_embed_spatial = torch.nn.Parameter(torch.zeros(1, 16, 8))
            self.pos_embed_temporal = torch.nn.Parameter(torch.zeros(1, T, 8))
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(2)])
            self.head = torch.nn.Linear(8, nout)

    ref_file = os.path.join(GOLDEN, "ref_checkpoint_epoch_00007.pyth")
    g = np.load(os.path.join(GOLDEN, "ref_checkpoint_loaded.npz"))
    dst = Tiny(8, 5)
    dst.load_state_dict({k: torch.from_numpy(g["before__" + k.replace(".", "__")]) for k in dst.state_dict()})
    not_loaded = []
    epoch = ck.load_checkpoint(ref_file, dst, epoch_reset=True, report=not_loaded)
    assert epoch == int(g["epoch"]) == -1
    assert sorted(not_loaded) == ["head.bias", "head.weight"]          # shape mismatch: left untouched, like the reference
    for k, v in dst.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(g[k.replace(".", "__")]), atol=1e-7), k
    # resume semantics: epoch and the torch AdamW state come back
    src = Tiny(4, 3)
    opt = torch.optim.AdamW(src.parameters(), lr=5e-4, eps=1e-8, weight_decay=0.05)
    not_loaded = []
    epoch = ck.load_checkpoint(ref_file, src, optimizer=opt, report=not_loaded)
    assert epoch == 6 and not not_loaded
    assert len(opt.state_dict()["state"]) == 8 and abs(opt.param_groups[0]["lr"] - 1e-3) < 1e-12
    # our writer -> same layout (keys, file name), readable by our reader
    cfg = load_yaml(YAML, ["NUM_GPUS", 0])
    path = ck.save_checkpoint(str(tmp_path), src, opt, 2, cfg)
    assert path.endswith(os.path.join("checkpoints", "checkpoint_epoch_00003.pyth")) and ck.has_checkpoint(str(tmp_path))
    assert ck.get_last_checkpoint(str(tmp_path)) == path
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(blob) == ["cfg", "epoch", "model_state", "optimizer_state"] and isinstance(blob["cfg"], str)
    again = Tiny(4, 3)
    assert ck.load_checkpoint(path, again) == 2
    assert all(torch.equal(a, b) for a, b in zip(again.state_dict().values(), src.state_dict().values()))


def test_mixed_precision_key_is_mapped_not_ignored(caplog):
    """TRAIN.MIXED_PRECISION (reference: fp16 autocast + GradScaler): with COMPUTE "auto" it selects the fp16 mode (the
    reference's own arithmetic, loss scaling included) and says so, the reference default (False) means fp32 arithmetic; an
    explicit COMPUTE wins and says so; the shipped YAMLs name bf16."""
    import logging
    from csts_amd.config import load_yaml
    from csts_amd.model import resolve_compute
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "CSTS_AMD.COMPUTE", "auto"])
    assert resolve_compute(cfg) == "fp32"
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "CSTS_AMD.COMPUTE", "auto", "TRAIN.MIXED_PRECISION", True])
    with caplog.at_level(logging.WARNING, logger="csts_amd"):
        assert resolve_compute(cfg) == "fp16"
    assert "MIXED_PRECISION" in caplog.text and "fp16" in caplog.text and "loss scaling" in caplog.text
    caplog.clear()
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "TRAIN.MIXED_PRECISION", True])          # the YAML's explicit bf16
    with caplog.at_level(logging.WARNING, logger="csts_amd"):
        assert resolve_compute(cfg) == "bf16"
    assert "bf16 compute mode" in caplog.text and "no loss scaling" in caplog.text
    caplog.clear()
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "CSTS_AMD.COMPUTE", "fp32", "TRAIN.MIXED_PRECISION", True])
    with caplog.at_level(logging.WARNING, logger="csts_amd"):
        assert resolve_compute(cfg) == "fp32"
    assert "explicit compute mode wins" in caplog.text
    assert resolve_compute(load_yaml(YAML, ["NUM_GPUS", 0])) == "bf16"      # the shipped YAMLs name their mode


def test_bench_gpus_n_without_launcher_never_measures_fewer_ranks():
    """`python bench.py --gpus 2` with no launcher and fewer than 2 GPUs: non-zero exit, nothing on stdout (it
    used to fall through to a one-rank run under an N-GPU label)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "refusing" in p.stderr


def test_act_checkpoint_key_is_logged_not_ignored(caplog):
    """MODEL.ACT_CHECKPOINT (custom_multimodal_builder.py:154,178,214: fairscale checkpoint_wrapper) changes memory, not
    values: accepted, and the decision is logged when the key is set."""
    import logging
    from csts_amd.config import load_yaml
    from csts_amd.registry import MODEL_REGISTRY
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "MODEL.ACT_CHECKPOINT", True])
    with caplog.at_level(logging.WARNING, logger="csts_amd"):
        MODEL_REGISTRY.get("CSTS")(cfg)
    assert any("ACT_CHECKPOINT" in r.message for r in caplog.records)


def test_video_and_audio_pretraining_pair_matches_reference_loader():
    """csts_amd.checkpoint.load_video_and_audio_checkpoints against what the REFERENCE's loader of the same name
    (slowfast/utils/checkpoint.py:357-470) made of two .pyth files its own save_checkpoint wrote (fixtures from
    oracle/gen_golden.py::gen_checkpoint_pair): audio entries win over video ones, all four position embeddings are resized from
    their own modality's file, shape mismatches stay untouched, epoch comes from the video file unless reset; and the
    TRAIN.CHECKPOINT_FILE_PATH + TRAIN.AUDIO_CHECKPOINT_FILE_PATH branch of load_train_checkpoint (:645-656) uses it."""
    from csts_amd import checkpoint as ck
    from csts_amd.config import load_yaml

    class Tiny(torch.nn.Module):
        def __init__(self, T, S, nout):
            super().__init__()
            self.pos_embed_spatial = torch.nn.Parameter(torch.zeros(1, S, 8))
            self.pos_embed_temporal = torch.nn.Parameter(torch.zeros(1, T, 8))
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(2)])
            self.pos_embed_spatial_audio = torch.nn.Parameter(torch.zeros(1, S, 8))
            self.pos_embed_temporal_audio = torch.nn.Parameter(torch.zeros(1, T, 8))
            self.blocks_audio = torch.nn.ModuleList([torch.nn.Linear(8, 8)])
            self.head = torch.nn.Linear(8, nout)

    g = np.load(os.path.join(GOLDEN, "ref_pretrain_pair_loaded.npz"))
    vid, aud = os.path.join(GOLDEN, "ref_pretrain_video.pyth"), os.path.join(GOLDEN, "ref_pretrain_audio.pyth")
    key = lambda k: k.replace(".", "__")
    dst = Tiny(8, 16, 5)
    dst.load_state_dict({k: torch.from_numpy(g["before__" + key(k)]) for k in dst.state_dict()})
    missing = []
    epoch = ck.load_video_and_audio_checkpoints(vid, aud, dst, epoch_reset=True, report=missing)
    assert epoch == int(g["epoch"]) == -1 and sorted(missing) == ["head.bias", "head.weight"]
    for k, v in dst.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(g[key(k)]), atol=1e-7), k
    # epoch from the VIDEO file, head taken from the audio file (it comes second), through load_train_checkpoint's branch
    dst2 = Tiny(8, 16, 3)
    dst2.load_state_dict({k: torch.from_numpy(g["before__" + key(k)]) for k in dst2.state_dict() if not k.startswith("head")}, strict=False)
    cfg = load_yaml(YAML, ["NUM_GPUS", 0, "TRAIN.AUTO_RESUME", False, "TRAIN.CHECKPOINT_FILE_PATH", vid,
                           "TRAIN.AUDIO_CHECKPOINT_FILE_PATH", aud, "TRAIN.CHECKPOINT_EPOCH_RESET", False])
    start = ck.load_train_checkpoint(cfg, dst2, None)
    assert start == int(g["epoch_no_reset"]) + 1 == 5
    for k, v in dst2.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(g["noreset__" + key(k)]), atol=1e-7), k
