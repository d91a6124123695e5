"""One rank of the two-process data-parallel GPU test (tests/test_gpu_dist.py starts it as a FRESH child process; it is not a
test module).  Every rank puts the real CSTS model on cuda:0, joins a gloo process group (RCCL refuses two ranks on one
device; gloo moves the same buffers through the host), and runs csts_amd.train.SegmentedTrainStep -- the HIP-graph chain with
eager collectives that bench.py times for N > 1 -- on ITS OWN clip of the seed-1000 B=2 batch that the reference fixture
tests/golden/model_T8_B2.npz was generated on (tools/train_avgaze_net.py:70-99 under DDP; slowfast/utils/distributed.py:15-49).

Step A (lr = 0: weights stay): what the optimizer graph reads through p.grad -- the averaged flat buckets -- is written out as
per-tensor norms + leading slices, with the loss terms.  Step B (lr = 1e-4, clip 1.0): one real update; a checksum of every
parameter is written so the parent can check that the replicas stayed bit-identical.

    python tests/dp_worker.py RANK WORLD PORT TRUNK_CUT COMPUTE OUT.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, trunk_cut, compute, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
    bucket_dtype = sys.argv[7] if len(sys.argv) > 7 else "fp32"
    factors = (sys.argv[8] != "dense") if len(sys.argv) > 8 else True
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from csts_amd.config import load_yaml
    from csts_amd.build import build_model
    from csts_amd import train as T, ops, distributed as du
    from oracle import csts_oracle as O               # test infrastructure: seeded weights + the fixture's batch

    cfg = load_yaml(os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"),
                    ["NUM_GPUS", 1, "MODEL.LOSS_FUNC", "kldiv+egonce", "DATA.NUM_FRAMES", 8, "CSTS_AMD.COMPUTE", compute,
                     "CSTS_AMD.TRUNK_CUT", trunk_cut, "CSTS_AMD.GRAD_BUCKET_DTYPE", bucket_dtype,
                     "CSTS_AMD.FUSION_GRAD_FACTORS", factors,
                     "CSTS_AMD.FACTORED_ADAMW", (len(sys.argv) > 9 and sys.argv[9] == "factored_adamw")])
    core = build_model(cfg)
    core.load_state_dict(O.seeded_params(8, 256), strict=True)
    core.eval()                                        # drop-path off: the fixture is an eval-mode forward + backward
    model = du.GradAllReduce(core, bucket_mb=cfg.CSTS_AMD.GRAD_BUCKET_MB)
    full = O.synthetic_batch(world, 8, 256, seed=1000)
    batch = {k: v[rank:rank + 1].contiguous().to(dev) for k, v in full.items() if k in ("video", "audio", "labels_hm")}
    opt = T.construct_optimizer(model, cfg, capturable=True)
    assert opt.max_grad_norm == 1.0                   # the YAML's clip; the kernels read the gradients, they never scale them in place
    step = T.SegmentedTrainStep(cfg, model, opt, batch, warmup=1)
    assert step.dist and step.use_graphs and step.trunk_cut == trunk_cut and not model.hooks_enabled
    assert set(step.graphs) == {"fwd", "loss", "bwd_head", "bwd_trunk", "opt"} | ({"bwd_trunk_early"} if trunk_cut else set())

    # ---- step A: gradients as the optimizer sees them
    loss, kld, nce = step.run(batch, lr=0.0)
    torch.cuda.synchronize()
    res = {"loss": float(loss), "kld": float(kld), "nce": float(nce), "n_buckets": len(step.flat), "n_factor_params": len(step.factor_params)}
    ptrs = [(f.data_ptr(), f.data_ptr() + f.numel() * 4) for f, _ in step.flat]
    names, norms, total = [], [], 0.0
    avg = step.averaged_grads()        # what the optimizer graph reads: views of the fp32 buckets (= p.grad) or of the 16-bit ones
    res["bucket_dtype"] = str(next(iter(avg.values())).dtype)
    for n, p in core.named_parameters():
        assert p.grad is not None and any(lo <= p.grad.data_ptr() < hi for lo, hi in ptrs), n      # views of the flat buckets
        assert bucket_dtype != "fp32" or avg[p].data_ptr() == p.grad.data_ptr()
        g = avg[p].double()
        names.append(n)
        norms.append(float(g.norm()))
        total += float((g * g).sum())
        res["g__" + n] = avg[p].flatten()[:64].float().cpu().numpy()
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array(norms)
    res["grad_total_norm"] = total ** 0.5
    res["weights_moved_by_lr0"] = float(sum((p.detach().double() - O.seeded_tensor(n, tuple(p.shape)).to(dev).double()).abs().sum()
                                            for n, p in list(core.named_parameters())[:8]))

    # ---- step B: one real update (clip + AdamW on the averaged buckets); replicas must stay identical
    res["clip_norm_seen"] = float(opt.grad_norm)       # total L2 norm the clip kernel measured on the averaged buckets
    opt.reset_state()
    loss_b, _, _ = step.run(batch, lr=1e-4)
    torch.cuda.synchronize()
    sums = [float(p.detach().double().sum()) for p in core.parameters()]
    absum = [float(p.detach().double().abs().sum()) for p in core.parameters()]
    res["param_sum"] = np.array(sums)
    res["param_abs_sum"] = np.array(absum)
    res["loss_b"] = float(loss_b)
    flat = torch.cat([p.detach().reshape(-1)[:256].float() for p in core.parameters()]).cpu().numpy()
    res["param_heads"] = flat
    np.savez(out, **res)
    ops.reset_deferred()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
