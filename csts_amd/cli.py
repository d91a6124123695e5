"""Command line + launcher: this package's counterpart of tools/run_net.py:11-25, slowfast/utils/parser.py:13-94,
slowfast/utils/misc.py:283-311 and slowfast/utils/multiprocessing.py:9-62.  Same flags (--cfg, --init_method,
--shard_id, --num_shards, trailing KEY VALUE opts); one process per GPU via torch.multiprocessing.spawn, RCCL
process group, then the train / test drivers below on SYNTHETIC clips (the dataset pipeline is out of scope)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

from .config import assert_and_infer_cfg, get_cfg


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="CSTS (MI355X) training and testing pipeline.")
    p.add_argument("--shard_id", default=0, type=int, help="The shard id of current node, starts from 0")
    p.add_argument("--num_shards", default=1, type=int, help="Number of shards used by the job")
    p.add_argument("--init_method", default="tcp://127.0.0.1:9999", type=str, help="TCP or shared file-system init")
    p.add_argument("--cfg", dest="cfg_file", default="configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml", type=str)
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE overrides")
    return p.parse_args(argv)


def load_config(args):
    cfg = get_cfg()
    if args.cfg_file is not None:
        cfg.merge_from_file(args.cfg_file)
    if args.opts:
        cfg.merge_from_list(args.opts)
    if hasattr(args, "num_shards") and hasattr(args, "shard_id"):
        cfg.NUM_SHARDS = args.num_shards
        cfg.SHARD_ID = args.shard_id
    os.makedirs(os.path.join(cfg.OUTPUT_DIR, "checkpoints"), exist_ok=True)
    return cfg


def _run(local_rank, num_proc, func, init_method, shard_id, num_shards, backend, cfg):
    world_size = num_proc * num_shards
    rank = shard_id * num_proc + local_rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    torch.distributed.init_process_group(backend=backend, init_method=init_method, world_size=world_size, rank=rank)
    try:
        func(cfg)
    finally:
        torch.distributed.destroy_process_group()


def launch_job(cfg, init_method, func, daemon=False):
    if cfg.NUM_GPUS > 1:
        torch.multiprocessing.spawn(_run, nprocs=cfg.NUM_GPUS,
                                    args=(cfg.NUM_GPUS, func, init_method, cfg.SHARD_ID, cfg.NUM_SHARDS, cfg.DIST_BACKEND, cfg),
                                    daemon=daemon)
    else:
        func(cfg=cfg)


def _is_master():
    return not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0


def _log(stats: dict):
    if _is_master():
        # six significant digits (a learning rate of 9.97e-5 must not print as 0.0001)
        print("json_stats: " + json.dumps({k: (float(f"{v:.6g}") if isinstance(v, float) else v) for k, v in stats.items()}),
              flush=True)


def is_checkpoint_epoch(cfg, cur_epoch: int) -> bool:
    """slowfast/utils/checkpoint.py:87-107 (no multigrid schedule on this path): the last epoch, and every CHECKPOINT_PERIOD-th."""
    return cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH or (cur_epoch + 1) % cfg.TRAIN.CHECKPOINT_PERIOD == 0


def is_eval_epoch(cfg, cur_epoch: int) -> bool:
    """slowfast/utils/misc.py:200-221: the last epoch, and every EVAL_PERIOD-th."""
    return cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH or (cur_epoch + 1) % cfg.TRAIN.EVAL_PERIOD == 0


@torch.no_grad()
def eval_epoch(cfg, model, cur_epoch: int, dev, rank: int, world: int):
    """tools/train_avgaze_net.py:158-219 on synthetic validation clips: eval-mode forward -> frame_softmax(T = 2) -> gather of
    predictions / heat-map labels / gaze labels over the ranks (:192-193) -> min-max rescale + adaptive_f1 on the device
    (:196-199) -> the epoch means the reference's ValGazeMeter logs (meters.py: f1 / recall / precision)."""
    from . import train as T
    from . import losses, metrics
    from . import distributed as du
    core = model.module if hasattr(model, "module") else model
    was_training = core.training
    model.eval()
    b = max(1, cfg.TRAIN.BATCH_SIZE // world)
    n = int(getattr(cfg.CSTS_AMD, "EVAL_STEPS", 2))
    acc = [0.0, 0.0, 0.0]
    for it in range(n):
        batch = T.synthetic_batch(b, cfg.DATA.NUM_FRAMES, cfg.DATA.TEST_CROP_SIZE, 500000 + rank + 7919 * it, dev)
        preds = losses.frame_softmax(model([batch["video"]], batch["audio"]), temperature=2)
        labels_hm, labels = batch["labels_hm"], batch["labels"]
        if world > 1:
            preds, labels_hm, labels = du.all_gather([preds, labels_hm, labels])
        f1, recall, precision, threshold = metrics.adaptive_f1(preds, labels_hm, labels, dataset=cfg.TRAIN.DATASET, rescale=True)
        acc = [a + v for a, v in zip(acc, (f1, recall, precision))]
    _log({"_type": "val_epoch", "epoch": cur_epoch + 1, "f1": acc[0] / n, "recall": acc[1] / n, "precision": acc[2] / n,
          "iters": n})
    if was_training:
        model.train()


def train(cfg):
    """Epoch loop of tools/train_avgaze_net.py:246-361 on synthetic clips."""
    from .build import build_model
    from . import train as T
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg)
    optimizer = T.construct_optimizer(model, cfg)
    world = max(cfg.NUM_GPUS, 1)
    b = cfg.TRAIN.BATCH_SIZE // world
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    dev = torch.device("cuda", torch.cuda.current_device())
    steps = cfg.CSTS_AMD.STEPS_PER_EPOCH
    from . import checkpoint as ck
    scaler = T.scaler_of(optimizer)        # fp16 mode: GradScaler state (train_avgaze_net.py:277), saved as "scaler_state"
    start_epoch = ck.load_train_checkpoint(cfg, model, optimizer, scaler=scaler)          # train_avgaze_net.py:280
    _log({"_type": "train_start", "start_epoch": start_epoch + 1, "resumed": start_epoch > 0,
          "optimizer_steps": int(optimizer.step_count()) if hasattr(optimizer, "step_count") else None})
    model.train()
    # the iteration runs from HIP graphs (the step is launch-bound from Python): one graph on a single GPU, a chain of graphs
    # with the RCCL collectives issued eagerly between them when data-parallel (train.SegmentedTrainStep)
    graphed = None
    # CSTS_AMD.EPOCHS_THIS_RUN > 0 ends this invocation after that many epochs (a pre-empted job, for resume tests); the
    # schedule and the checkpoint / eval periods still follow SOLVER.MAX_EPOCH
    last = cfg.SOLVER.MAX_EPOCH
    if int(getattr(cfg.CSTS_AMD, "EPOCHS_THIS_RUN", 0) or 0) > 0:
        last = min(last, start_epoch + int(cfg.CSTS_AMD.EPOCHS_THIS_RUN))
    for epoch in range(start_epoch, last):
        t0 = time.time()
        for it in range(steps):
            batch = T.synthetic_batch(b, cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE, 1000 + rank + 7919 * (epoch * steps + it), dev)
            lr = T.get_lr_at_epoch(cfg, epoch + float(it) / steps)
            if getattr(cfg.CSTS_AMD, "HIP_GRAPH", True):
                if graphed is None:
                    graphed = (T.GraphedTrainStep if world == 1 else T.SegmentedTrainStep)(cfg, model, optimizer, batch)
                loss, kld, nce = graphed.run(batch, lr)
            else:
                loss, kld, nce = T.train_step(cfg, model, batch, optimizer, lr)
            if (it + 1) % cfg.LOG_PERIOD == 0:
                vals = T.du.all_reduce([loss, kld] + ([nce] if nce is not None else []))
                lv = float(vals[0])
                if not (lv == lv) or lv in (float("inf"), float("-inf")):
                    raise RuntimeError("ERROR: Got NaN losses")      # misc.check_nan_losses (misc.py:26-33)
                _log({"_type": "train_iter", "epoch": epoch + 1, "iter": it + 1, "lr": lr,
                      "lr_device": float(optimizer.param_groups[0]["lr"]),        # what the (captured) optimizer kernels read
                      "loss_scale": scaler.get_scale() if scaler is not None else None,
                      "loss": lv,
                      "kldiv_loss": float(vals[1]), "nce_loss": float(vals[2]) if nce is not None else None})
        torch.cuda.synchronize()
        _log({"_type": "train_epoch", "epoch": epoch + 1, "clips_per_s": steps * b * world / (time.time() - t0)})
        if getattr(cfg.CSTS_AMD, "SAVE_CHECKPOINTS", False) and is_checkpoint_epoch(cfg, epoch):
            path = ck.save_checkpoint(cfg.OUTPUT_DIR, model, optimizer, epoch, cfg, scaler=scaler)  # train_avgaze_net.py:337-346 (0.75 GB + moments)
            _log({"_type": "checkpoint", "epoch": epoch + 1, "path": path,
                  "optimizer_steps": int(optimizer.step_count()) if hasattr(optimizer, "step_count") else None})
        if is_eval_epoch(cfg, epoch):                                         # train_avgaze_net.py:338,355-356
            eval_epoch(cfg, model, epoch, dev, rank, world)


@torch.no_grad()
def test(cfg):
    """Forward-only driver (tools/test_avgaze_net.py:21-141) on synthetic clips: model(inputs, audio) -> frame_softmax."""
    from .build import build_model
    from . import train as T
    from . import losses
    model = build_model(cfg)
    model.eval()
    world = max(cfg.NUM_GPUS, 1)
    b = max(1, min(cfg.TEST.BATCH_SIZE // world, 8))
    dev = torch.device("cuda", torch.cuda.current_device())
    batch = T.synthetic_batch(b, cfg.DATA.NUM_FRAMES, cfg.DATA.TEST_CROP_SIZE, 2000, dev)
    preds = losses.frame_softmax(model([batch["video"]], batch["audio"]), temperature=2)
    # tools/test_avgaze_net.py:66-69: min-max rescale per frame, then the adaptive-threshold F1 -- both on the device
    from . import metrics
    f1, recall, precision, threshold = metrics.adaptive_f1(preds, batch["labels_hm"], batch["labels"],
                                                           dataset=cfg.TEST.DATASET, rescale=True)
    _log({"_type": "test", "preds_shape": list(preds.shape), "preds_sum": float(preds.sum()), "f1": f1, "recall": recall,
          "precision": precision, "threshold": float(threshold)})


def main(argv=None):
    args = parse_args(argv)
    if len(sys.argv) == 1 and argv is None:
        print("usage: run_net.py --cfg <yaml> [KEY VALUE ...]")
    cfg = load_config(args)
    cfg = assert_and_infer_cfg(cfg)
    if cfg.TRAIN.ENABLE:
        launch_job(cfg=cfg, init_method=args.init_method, func=train)
    if cfg.TEST.ENABLE:
        launch_job(cfg=cfg, init_method=args.init_method, func=test)
