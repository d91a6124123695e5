#!/usr/bin/env python3
"""bench.py -- CSTS training throughput on MI355X (BASELINE.json metric: clips/s, training step).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): CSTS_Ego4D_Gaze_Forecast.yaml + MODEL.LOSS_FUNC kldiv+egonce, 16 frames x 256^2
(the reference itself cannot run 224^2), b = 4 clips per GPU (weak scaling), bf16 MFMA mode with fp32
residual stream, synthetic clips + 24 kHz STFT resident in HBM before the timed region.  One step = forward +
KLDiv + 0.05 EgoNCE + backward (+ bucketed RCCL gradient all-reduce when N > 1) + L2 clip + AdamW: nothing of the
reference iteration (train_avgaze_net.py:65-109) is skipped.  Rank 0 prints ONE JSON line.

Extra objects: "roofline" for the dominant kernel (the bf16 NT GEMM), measured live with HIP events on the
launch stream over an instrumented pass, and "cpu_baseline" = the CPU oracle (a port, plain PyTorch fp32) timed on
the host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FWD_GFLOP_PER_CLIP = {8: 208.7, 16: 465.3, 32: 1122.3}      # SURVEY.md 8(d) / BASELINE.md section 2
BYTES_FWD_GB_PER_CLIP = {8: 1.24, 16: 2.34, 32: 4.53}
PEAK_BF16_TFLOPS = 2500.0                                     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--frames", type=int, default=16)
    p.add_argument("--batch-per-gpu", type=int, default=4)
    p.add_argument("--compute", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--rehearse-dist", action="store_true",
                   help="run the N>1 code path (RCCL process group, GradAllReduce buckets, eager step) with ONE rank")
    p.add_argument("--ddp-graph", action="store_true",
                   help="EXPERIMENTAL: capture the data-parallel step (RCCL collectives included) into a HIP graph")
    p.add_argument("--no-graph", action="store_true", help="do not capture the step into a HIP graph (single GPU only)")
    p.add_argument("--op-breakdown", default=None, help="write per-C-ABI-entry device time of one eager step to this file")
    p.add_argument("--dump-gemm", default=None, help="write a per-shape GEMM timing table to this file")
    return p.parse_args()


class GemmTimer:
    """HIP-event timing of every csts_gemm call at the C-ABI boundary (events recorded on the launch stream = torch's
    current stream).  Each call is attributed to the kernel csts_gemm picks for it (csts_gemm_plan), named exactly as
    rocprofv3 prints it, so the live numbers can be checked against profiles/*_kernel_stats.csv."""

    class _Proxy:
        def __init__(self, lib, owner):
            self._lib, self._owner = lib, owner

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if name != "csts_gemm":
                return fn
            owner, lib = self._owner, self._lib

            def timed(argsref, stream):
                import ctypes as C
                a = argsref._obj
                v2, tr, ns = C.c_int(), C.c_int(), C.c_int()
                lib.csts_gemm_plan(argsref, C.byref(v2), C.byref(tr), C.byref(ns))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(argsref, stream)
                e1.record()
                tf = lambda b: "true" if b else "false"
                if v2.value >= 30:
                    name = f"gemm3_kernel<{tr.value // 64}, {v2.value - 30}>"
                elif v2.value:
                    name = (f"gemm2_kernel<{tf(a.layout != 2)}, {tf(a.layout == 0)}, {tf(a.a_dt == 0)}, {tf(a.b_dt == 0)}, "
                            f"{tr.value // 64}, 2>")
                else:
                    name = f"gemm_kernel<{tf(a.layout != 2)}, {tf(a.layout == 0)}, {tf(a.compute == 0)}>"
                esz = lambda dt: 4 if dt == 0 else 2
                byt = a.M * a.K * esz(a.a_dt) + a.N * a.K * esz(a.b_dt) + a.M * a.N * esz(a.c_dt)
                owner.records.append((name, ns.value, 2.0 * a.M * a.N * a.K, byt, e0, e1, (a.layout, a.M, a.N, a.K, ns.value)))
                return rc
            return timed

    def __init__(self):
        self.records = []   # (kernel name, nsplit, flops, bytes, ev0, ev1, shape)

    def install(self):
        from csts_amd import lib as L
        self._orig = L.load()
        L._lib = GemmTimer._Proxy(self._orig, self)

    def remove(self):
        from csts_amd import lib as L
        L._lib = self._orig

    def summary(self):
        """kernel name -> [launches, flops, bytes, seconds, has_finish_pass]"""
        torch.cuda.synchronize()
        agg = {}
        for name, ns, fl, by, e0, e1, _shape in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, False])
            a[0] += 1
            a[1] += fl
            a[2] += by
            a[3] += e0.elapsed_time(e1) * 1e-3
            a[4] = a[4] or ns > 1
        return agg

    def dump_shapes(self, path):
        torch.cuda.synchronize()
        per = {}
        for name, ns, fl, by, e0, e1, shape in self.records:
            a = per.setdefault(shape, [0, 0.0, fl, by])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
        rows = sorted(per.items(), key=lambda kv: -kv[1][1])
        with open(path, "w") as f:
            f.write("layout M N K split calls total_ms avg_us TFLOPs GBps\n")
            for (lay, M, N, K, sp), (n, sec, fl, by) in rows:
                f.write(f"{'NT NN TN'.split()[lay]} {M} {N} {K} {sp} {n} {sec*1e3:.3f} {sec/n*1e6:.1f} {fl*n/sec/1e12:.1f} {by*n/sec/1e9:.0f}\n")


class OpTimer:
    """HIP-event timing of EVERY C-ABI entry point (per-family device time of one eager single-stream step)."""

    class _Proxy:
        def __init__(self, lib, rec):
            self._lib, self._rec = lib, rec

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if not name.startswith("csts_") or name.endswith("_workspace") or name in (
                    "csts_last_error", "csts_abi_version", "csts_gemm_v2_eligible"):
                return fn
            rec = self._rec

            def timed(*a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*a)
                e1.record()
                rec.append((name, e0, e1))
                return rc
            return timed

    def __init__(self):
        self.rec = []

    def install(self):
        from csts_amd import lib as L
        self._orig = L.load()
        L._lib = OpTimer._Proxy(self._orig, self.rec)

    def remove(self):
        from csts_amd import lib as L
        L._lib = self._orig

    def summary(self, steps):
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1 in self.rec:
            a = agg.setdefault(name[5:], [0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
        return {k: {"calls": v[0] // steps, "ms": round(v[1] / steps, 3)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}


def cpu_baseline(frames):
    """Oracle (CPU port, fp32 PyTorch ops) fwd + loss + bwd on ONE clip of the same shape; a bounded sample."""
    from oracle import csts_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))   # more threads only add contention for these op sizes
    P = {k: v.requires_grad_(True) for k, v in O.seeded_params(frames, 256).items()}
    batch = O.synthetic_batch(2, frames, 256, seed=1000)    # B=2: EgoNCE is identically 0 at B=1
    t0 = time.time()
    logits, v, a = O.csts_forward(P, batch["video"], batch["audio"], frames, 256, return_embed=True)
    loss, _, _ = O.csts_loss(logits, v, a, batch["labels_hm"], 0.05)
    loss.backward()
    dt = time.time() - t0
    return {"value": round(2 / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 un-warmed train step (fwd+loss+bwd, no optimizer) of the CPU oracle, B=2, {frames}x256^2, fp32, {dt:.1f} s"}


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): native libraries print there too (RCCL writes a five-line version banner
    # to stdout when its first communicator comes up), so fd 1 points at stderr until the result is printed
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_path = world > 1 or args.rehearse_dist
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        if world == 1:
            torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
            from csts_amd import distributed as _du
            _du._FORCE = True        # 1-rank rehearsal: take the collective code paths anyway
        else:
            torch.distributed.init_process_group(backend="nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    from csts_amd import train as T
    from csts_amd.distributed import GradAllReduce

    b = args.batch_per_gpu
    cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"),
                    ["NUM_GPUS", world, "TRAIN.BATCH_SIZE", b * world, "MODEL.LOSS_FUNC", "kldiv+egonce",
                     "MODEL.LOSS_ALPHA", 0.05, "DATA.NUM_FRAMES", args.frames, "CSTS_AMD.COMPUTE", args.compute])
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg)
    if dist_path and not isinstance(model, GradAllReduce):
        model = GradAllReduce(model, bucket_mb=cfg.CSTS_AMD.GRAD_BUCKET_MB)
    model.train()
    use_graph = (not dist_path or args.ddp_graph) and not args.no_graph
    opt = T.construct_optimizer(model, cfg, capturable=use_graph)
    batch = T.synthetic_batch(b, args.frames, 256, 1000 + rank, dev)      # resident in HBM before timing
    lr = T.get_lr_at_epoch(cfg, 0.0)
    graphed = T.GraphedTrainStep(cfg, model, opt, batch, allow_collectives=args.ddp_graph) if use_graph else None

    def step():
        if graphed is not None:
            return graphed.run(batch, lr)
        return T.train_step(cfg, model, batch, opt, lr)

    def eager_step():
        return T.train_step(cfg, model, batch, opt, lr)

    def sync():
        torch.cuda.synchronize()
        if dist_path:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, kld, nce = step()
    sync()
    dt = time.perf_counter() - t0
    if dist_path:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    clips = b * world * args.steps
    value = clips / dt
    ms_per_step = dt / args.steps * 1e3

    roof = None
    if not args.no_roofline:
        # every rank runs the instrumented steps (they contain the gradient collectives); rank 0 reports its own timings
        from csts_amd import ops as _ops
        _mode = _ops.GROUP_WGRADS
        if use_graph and _mode == "capture":
            _ops.GROUP_WGRADS = "always"      # instrument the same kernel set the captured (timed) step runs
        gt = GemmTimer()
        gt.install()
        core = model.module if hasattr(model, "module") else model
        two = core.two_streams
        core.two_streams = False        # one stream: an event pair then brackets exactly one kernel (plus its launch gap)
        # keep the launch stream ahead of the GPU: a spin kernel first, so that the host has every launch of the step
        # queued before the GPU reaches it and an event pair brackets the kernel alone, not the host's launch gap
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
        spin = int(10_000_000 * 120.0 / max(e0.elapsed_time(e1), 1e-3))     # ~120 ms head start
        for _ in range(2):
            torch.cuda._sleep(spin)
            eager_step()
        gt.remove()
        if args.op_breakdown:
            ot = OpTimer()
            ot.install()
            torch.cuda._sleep(spin)
            eager_step()
            t_ev0, t_ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ot.rec.clear()
            torch.cuda._sleep(spin)
            t_ev0.record()
            eager_step()
            t_ev1.record()
            ot.remove()
            ops_ms = ot.summary(1)
            tot = t_ev0.elapsed_time(t_ev1)
            with open(args.op_breakdown if rank == 0 else os.devnull, "w") as f:
                f.write(f"single-stream eager step {tot:.2f} ms; C-ABI kernels {sum(v['ms'] for v in ops_ms.values()):.2f} ms; "
                        f"torch-native remainder (optimizer, clip, autograd adds, casts, fills) {tot - sum(v['ms'] for v in ops_ms.values()):.2f} ms\n")
                for k, v in ops_ms.items():
                    f.write(f"{k:28s} {v['calls']:5d} calls {v['ms']:8.3f} ms\n")
        core.two_streams = two
        _ops.GROUP_WGRADS = _mode
        agg = gt.summary()
        if args.dump_gemm and rank == 0:
            gt.dump_shapes(args.dump_gemm)
        # dominant kernel = the single-kernel GEMM variant (no split-K finishing pass inside the event pair) with the most time
        single = {k: v for k, v in agg.items() if not v[4]}
        name = max(single, key=lambda k: single[k][3]) if single else None
        n, fl, by, sec, _ = agg.get(name, (0, 0.0, 0.0, 1.0, False))
        tot_sec = sum(a[3] for a in agg.values())
        tot_fl = sum(a[1] for a in agg.values())
        peak = PEAK_BF16_TFLOPS if args.compute == "bf16" else 157.3
        traffic = None
        try:     # HBM bytes per launch from the rocprofv3 --pmc passes (tools/pmc_traffic.py), same command, same kernel
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if name in pmc["kernels"]:
                traffic = pmc["kernels"][name]["hbm_bytes_per_launch"]
        except Exception:
            pass
        # the binding roofline of this kernel: t_min = max(flops / P_mfma, algorithmic bytes / P_hbm)
        t_mfma, t_hbm = fl / (peak * 1e12), by / (PEAK_HBM_GBS * 1e9)
        hbm_bound = t_hbm > t_mfma
        roof = {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
                "achieved": round(by / sec / 1e9, 1) if hbm_bound else round(fl / sec / 1e12, 2),
                "peak": PEAK_HBM_GBS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round(max(t_mfma, t_hbm) / sec, 4),
                "mfma_tflops": round(fl / sec / 1e12, 2), "mfma_frac": round(t_mfma / sec, 4),
                "algorithmic_gbps": round(by / sec / 1e9, 1), "hbm_frac": round(t_hbm / sec, 4),
                "traffic": traffic, "launches_per_step": n // 2, "avg_launch_us": round(sec / max(n, 1) * 1e6, 2),
                "algorithmic_flop_per_launch": round(fl / max(n, 1)), "algorithmic_bytes_per_launch": round(by / max(n, 1)),
                "all_gemm_tflops": round(tot_fl / tot_sec / 1e12, 2), "all_gemm_ms_per_step": round(tot_sec / 2 * 1e3, 2),
                "per_kernel": {k: {"launches_per_step": v[0] // 2, "avg_us": round(v[3] / v[0] * 1e6, 1),
                                   "tflops": round(v[1] / v[3] / 1e12, 1), "ms_per_step": round(v[3] / 2 * 1e3, 3)}
                               for k, v in sorted(agg.items(), key=lambda kv: -kv[1][3])[:8]}}
    if dist_path:
        torch.distributed.barrier()

    if rank == 0:
        per_gpu = value / world
        train_gflop = 3 * FWD_GFLOP_PER_CLIP.get(args.frames, 0.0)
        train_gb = 3 * BYTES_FWD_GB_PER_CLIP.get(args.frames, 0.0)
        out = {
            "metric": "clips/sec training (fwd+bwd) CSTS-Ego4D 16x256^2 bf16", "value": round(value, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.compute, "data": "synthetic",
            "config": {"workload": f"CSTS_Ego4D_Gaze_Forecast.yaml train step (fwd + KLDiv + 0.05*EgoNCE + bwd"
                                   f"{' + RCCL grad all-reduce' if dist_path else ''} + clip + AdamW), "
                                   f"{args.frames}x256^2 video + 24 kHz STFT audio, b={b}/GPU",
                       "global_batch": b * world, "frames": args.frames, "crop": 256, "parallelism": f"dp{world}", "hip_graph": bool(use_graph),
                       "note": "256^2 not 224^2: the reference's (1,8,8) fusion convs reject 224^2 (SURVEY.md D1)"},
            "loss": round(float(loss), 5),
            "end_to_end_roofline": {"mfma_frac": round(per_gpu * train_gflop / 1e3 / PEAK_BF16_TFLOPS, 4),
                                    "hbm_frac": round(per_gpu * train_gb / PEAK_HBM_GBS, 4),
                                    "train_gflop_per_clip": train_gflop, "train_gb_per_clip": train_gb},
        }
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.frames)
            except Exception as e:  # keep the line valid even if the host runs out of memory
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        sys.stdout.flush()
        os.dup2(_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)       # teardown chatter of native libraries stays off stdout as well
    if dist_path:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
