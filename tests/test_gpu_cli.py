"""The advertised entry point on the GPU (north_star: "driven by the existing tools/run_net.py + YAML configs"):
`python tools/run_net.py --cfg configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml ...` in fresh child processes -- train one epoch
from a HIP graph, checkpoint in the reference's wire format, periodic eval with adaptive F1, then a SECOND invocation that
auto-resumes (tools/run_net.py:11-25, tools/train_avgaze_net.py:246-361, slowfast/utils/checkpoint.py:617-659)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 4


def _run_net(out_dir, extra=(), mp=("TRAIN.MIXED_PRECISION", "True")):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_net.py"), "--cfg", os.path.join(ROOT, "configs/Ego4D/CSTS_Ego4D_Gaze_Forecast.yaml"),
           "NUM_GPUS", "1", "TRAIN.BATCH_SIZE", "4", "MODEL.LOSS_FUNC", "kldiv+egonce", "MODEL.LOSS_ALPHA", "0.05",
           *mp, "CSTS_AMD.STEPS_PER_EPOCH", str(STEPS), "CSTS_AMD.SAVE_CHECKPOINTS", "True",
           "TRAIN.CHECKPOINT_PERIOD", "1", "TRAIN.EVAL_PERIOD", "1", "CSTS_AMD.EPOCHS_THIS_RUN", "1", "LOG_PERIOD", "1",
           "OUTPUT_DIR", str(out_dir)] + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    recs = [json.loads(ln[len("json_stats: "):]) for ln in p.stdout.splitlines() if ln.startswith("json_stats: ")]
    assert recs, p.stdout[-2000:]
    return recs, p.stderr


def _of(recs, kind):
    return [r for r in recs if r["_type"] == kind]


def test_run_net_train_checkpoint_resume_eval(tmp_path):
    gold = np.load(os.path.join(GOLDEN, "lr_schedule.npz"))
    lr_at = dict(zip([float(e) for e in gold["epochs"]], [float(v) for v in gold["lr"]]))      # the reference's lr_policy (MAX_EPOCH 15)

    # ---- first invocation: epoch 1 of 15, TEST.ENABLE False
    recs, err = _run_net(tmp_path, ["TEST.ENABLE", "False"])
    start = _of(recs, "train_start")[0]
    assert start["start_epoch"] == 1 and start["resumed"] is False and start["optimizer_steps"] == 0
    assert "bf16 compute mode" in err                   # TRAIN.MIXED_PRECISION is never mapped silently
    iters = _of(recs, "train_iter")
    assert [r["iter"] for r in iters] == list(range(1, STEPS + 1)) and all(r["epoch"] == 1 for r in iters)
    for r in iters:
        assert np.isfinite(r["loss"]) and np.isfinite(r["kldiv_loss"]) and np.isfinite(r["nce_loss"])
        assert abs(r["loss"] - (r["kldiv_loss"] + 0.05 * r["nce_loss"])) < 1e-4
        assert abs(r["lr_device"] - r["lr"]) <= 2e-6 * r["lr"]          # the captured optimizer kernels read the schedule's value
    # cosine schedule at epoch 0.0 and 0.5 against the values the reference's lr_policy produced (tests/golden/lr_schedule.npz)
    assert abs(iters[0]["lr"] - lr_at[0.0]) <= 2e-6 * lr_at[0.0]
    assert abs(iters[STEPS // 2]["lr"] - lr_at[0.5]) <= 2e-6 * lr_at[0.5]
    assert iters[-1]["loss"] < iters[0]["loss"] + 0.5                     # AdamW at 1e-4 does not blow the loss up
    ck = _of(recs, "checkpoint")
    assert len(ck) == 1 and ck[0]["epoch"] == 1 and ck[0]["optimizer_steps"] == STEPS
    path = os.path.join(str(tmp_path), "checkpoints", "checkpoint_epoch_00001.pyth")
    assert ck[0]["path"] == path and os.path.exists(path)
    val = _of(recs, "val_epoch")
    assert len(val) == 1 and 0.0 <= val[0]["f1"] <= 1.0 and 0.0 <= val[0]["recall"] <= 1.0 and 0.0 <= val[0]["precision"] <= 1.0

    # ---- the file is the reference's wire format (slowfast/utils/checkpoint.py:110-143): keys, 524 named tensors with the
    # reference's names / shapes / order, torch.optim.AdamW-layout optimizer state with the step count
    sd = torch.load(path, map_location="cpu", weights_only=False)
    assert sd["epoch"] == 0 and set(sd) >= {"epoch", "model_state", "optimizer_state", "cfg"}
    manifest = json.load(open(os.path.join(GOLDEN, "manifest_T8.json")))["entries"]
    assert [(k, list(v.shape)) for k, v in sd["model_state"].items()] == [(n, list(s)) for n, s in manifest]
    ost = sd["optimizer_state"]
    assert len(ost["state"]) == len(manifest) and float(ost["state"][0]["step"]) == STEPS
    first_w = sd["model_state"]["blocks.3.mlp.fc1.weight"].clone()
    del sd

    # ---- second invocation: auto-resume (AUTO_RESUME True in the YAML), epoch 2, then the TEST.ENABLE pass
    recs2, _ = _run_net(tmp_path, ["TEST.ENABLE", "True"])
    start2 = _of(recs2, "train_start")[0]
    assert start2["start_epoch"] == 2 and start2["resumed"] is True and start2["optimizer_steps"] == STEPS
    iters2 = _of(recs2, "train_iter")
    assert all(r["epoch"] == 2 for r in iters2) and len(iters2) == STEPS
    assert abs(iters2[0]["lr"] - lr_at[1.0]) <= 2e-6 * lr_at[1.0]
    assert all(abs(r["lr_device"] - r["lr"]) <= 2e-6 * r["lr"] for r in iters2)
    ck2 = _of(recs2, "checkpoint")
    assert len(ck2) == 1 and ck2[0]["epoch"] == 2 and ck2[0]["optimizer_steps"] == 2 * STEPS
    assert len(_of(recs2, "val_epoch")) == 1
    tst = _of(recs2, "test")
    assert len(tst) == 1 and 0.0 <= tst[0]["f1"] <= 1.0 and tst[0]["preds_shape"][1:] == [1, 8, 64, 64]
    assert abs(tst[0]["preds_sum"] - tst[0]["preds_shape"][0] * 8) < 1e-2          # frame_softmax: every frame sums to 1
    sd2 = torch.load(os.path.join(str(tmp_path), "checkpoints", "checkpoint_epoch_00002.pyth"), map_location="cpu", weights_only=False)
    assert sd2["epoch"] == 1 and float(sd2["optimizer_state"]["state"][0]["step"]) == 2 * STEPS
    assert not torch.equal(sd2["model_state"]["blocks.3.mlp.fc1.weight"], first_w)      # training continued from the checkpoint
    for f in os.listdir(os.path.join(str(tmp_path), "checkpoints")):                    # 2.3 GB each
        os.remove(os.path.join(str(tmp_path), "checkpoints", f))


def test_run_net_fp16_checkpoint_carries_scaler_state(tmp_path):
    """CSTS_AMD.COMPUTE fp16 through the entry point: the loss scale is logged, the checkpoint holds "scaler_state" in
    torch.cuda.amp.GradScaler's layout (slowfast/utils/checkpoint.py:133-134), and the resumed run starts from it."""
    fp16 = ("TRAIN.MIXED_PRECISION", "True", "CSTS_AMD.COMPUTE", "fp16")
    recs, err = _run_net(tmp_path, ["TEST.ENABLE", "False"], mp=fp16)
    assert "fp16 autocast + GradScaler arithmetic" in err
    iters = _of(recs, "train_iter")
    assert len(iters) == STEPS and all(np.isfinite(r["loss"]) and r["loss_scale"] >= 1024.0 for r in iters)
    path = os.path.join(str(tmp_path), "checkpoints", "checkpoint_epoch_00001.pyth")
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sc = sd["scaler_state"]
    assert set(sc) >= {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    assert sc["scale"] == iters[-1]["loss_scale"] or sc["scale"] in (0.5 * iters[-1]["loss_scale"], 2.0 * iters[-1]["loss_scale"])
    assert sc["growth_factor"] == 2.0 and sc["backoff_factor"] == 0.5 and sc["growth_interval"] == 2000
    del sd
    recs2, _ = _run_net(tmp_path, ["TEST.ENABLE", "False"], mp=fp16)
    st = _of(recs2, "train_start")[0]
    assert st["resumed"] is True and st["start_epoch"] == 2
    it2 = _of(recs2, "train_iter")
    assert it2[0]["loss_scale"] == sc["scale"] and all(np.isfinite(r["loss"]) for r in it2)
    for f in os.listdir(os.path.join(str(tmp_path), "checkpoints")):
        os.remove(os.path.join(str(tmp_path), "checkpoints", f))
