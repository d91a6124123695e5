"""Checkpoint wire format of the reference (slowfast/utils/checkpoint.py): ``checkpoints/checkpoint_epoch_{:05d}.pyth`` =
``torch.save({"epoch", "model_state", "optimizer_state", "cfg"})`` written by the master process; loading matches tensors
by NAME AND SHAPE (fine-tuning from a Kinetics MViT-B or from released CSTS weights, :312-319) and bilinearly resizes
``pos_embed_spatial`` / ``pos_embed_temporal`` when only their length differs (:327-335).  I/O plumbing around the hot path
(SURVEY.md 8(f) rank 4): plain torch on the host, no kernels."""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn.functional as F
import yaml


def get_checkpoint_dir(path_to_job):
    """checkpoint.py:40-46."""
    return os.path.join(path_to_job, "checkpoints")


def get_path_to_checkpoint(path_to_job, epoch):
    """checkpoint.py:49-57."""
    return os.path.join(get_checkpoint_dir(path_to_job), "checkpoint_epoch_{:05d}.pyth".format(epoch))


def has_checkpoint(path_to_job):
    """checkpoint.py:76-84."""
    d = get_checkpoint_dir(path_to_job)
    return os.path.isdir(d) and any("checkpoint" in f for f in os.listdir(d))


def get_last_checkpoint(path_to_job):
    """checkpoint.py:60-73: the lexicographically last checkpoint file."""
    d = get_checkpoint_dir(path_to_job)
    names = sorted(f for f in (os.listdir(d) if os.path.isdir(d) else []) if "checkpoint" in f)
    assert len(names), "No checkpoints found in '{}'.".format(d)
    return os.path.join(d, names[-1])


def _core(model):
    return model.module if hasattr(model, "module") else model


def _is_master():
    return not (torch.distributed.is_available() and torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg, scaler=None):
    """checkpoint.py:110-143 (master process only; the data-parallel wrapper is stripped)."""
    if not _is_master():
        return None
    os.makedirs(get_checkpoint_dir(path_to_job), exist_ok=True)
    sd = OrderedDict((k, v.detach().cpu()) for k, v in _core(model).state_dict().items())
    checkpoint = {"epoch": epoch, "model_state": sd, "optimizer_state": optimizer.state_dict(),
                  "cfg": yaml.safe_dump(_plain(cfg))}
    if scaler is not None:
        checkpoint["scaler_state"] = scaler.state_dict()
    path = get_path_to_checkpoint(path_to_job, epoch + 1)
    with open(path, "wb") as f:
        torch.save(checkpoint, f)
    return path


def _plain(cfg):
    if isinstance(cfg, dict):
        return {k: _plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_plain(v) for v in cfg]
    return cfg


def load_checkpoint(path_to_checkpoint, model, data_parallel=False, optimizer=None, scaler=None, epoch_reset=False,
                    clear_name_pattern=(), report=None):
    """checkpoint.py:185-354 for the PyTorch (.pyth) branch: name + shape matching, pos-embed interpolation, optional
    optimizer state.  Returns the checkpoint's epoch (-1 when fine-tuning / epoch_reset), like the reference; pass a list
    as `report` to receive the names of the model entries the checkpoint did not provide (the reference only logs them)."""
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    ms = _core(model) if data_parallel or hasattr(model, "module") else model
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu", weights_only=False)
    pre = checkpoint["model_state"]
    for item in clear_name_pattern:                                   # :300-310
        pre = OrderedDict((k.replace(item, "") if item in k else k, v) for k, v in pre.items())
    model_dict = ms.state_dict()
    match = {k: v for k, v in pre.items() if k in model_dict and v.size() == model_dict[k].size()}     # :312-319
    not_loaded = [k for k in model_dict if k not in match]
    for k in ("pos_embed_spatial", "pos_embed_temporal"):             # :327-335 (the first two dims act as batch, channel)
        if k in not_loaded and k in pre:
            t = model_dict[k].size()
            match[k] = F.interpolate(pre[k].float().unsqueeze(0), (t[1], t[2]), mode="bilinear").squeeze(0)
            not_loaded.remove(k)
    ms.load_state_dict(match, strict=False)
    if hasattr(ms, "_refresh_w16") and next(ms.parameters()).is_cuda:
        ms._refresh_w16()                                             # bf16 shadows follow the new masters
    epoch = -1
    if "epoch" in checkpoint and not epoch_reset:                     # :346-353
        epoch = checkpoint["epoch"]
        if optimizer is not None:
            optimizer.load_state_dict(checkpoint["optimizer_state"])
        if scaler is not None and "scaler_state" in checkpoint:
            scaler.load_state_dict(checkpoint["scaler_state"])
    if report is not None:
        report.extend(not_loaded)
    return epoch


def load_video_and_audio_checkpoints(path_to_video_checkpoint, path_to_audio_checkpoint, model, data_parallel=False, optimizer=None,
                                     scaler=None, epoch_reset=False, clear_name_pattern=(), report=None):
    """checkpoint.py:357-470: initialise from a VIDEO pre-training checkpoint and an AUDIO one (TRAIN.CHECKPOINT_FILE_PATH +
    TRAIN.AUDIO_CHECKPOINT_FILE_PATH).  Entries are taken by name + shape, the audio file winning where both provide a name;
    the four position embeddings are resized bilinearly (first two dims as batch / channel) -- `pos_embed_spatial/temporal` from
    the video file, the `*_audio` twins from the audio file, whenever either file holds the name (the reference indexes the
    modality's own file there: a name held only by the other file raises KeyError, here too); epoch / optimizer / scaler state
    come from the VIDEO file unless epoch_reset.  Returns the epoch (-1 when reset / absent)."""
    for p_ in (path_to_video_checkpoint, path_to_audio_checkpoint):
        assert os.path.exists(p_), "Checkpoint '{}' not found".format(p_)
    ms = _core(model) if data_parallel or hasattr(model, "module") else model
    states = []
    for p_ in (path_to_video_checkpoint, path_to_audio_checkpoint):
        with open(p_, "rb") as f:
            states.append(torch.load(f, map_location="cpu", weights_only=False))
    video_ck, audio_ck = states
    pre = []
    for ck in states:
        sd = ck["model_state"]
        for item in clear_name_pattern:
            sd = OrderedDict((k.replace(item, "") if item in k else k, v) for k, v in sd.items())
        pre.append(sd)
    video_sd, audio_sd = pre
    model_dict = ms.state_dict()
    match = {k: v for k, v in video_sd.items() if k in model_dict and v.size() == model_dict[k].size()}
    match.update({k: v for k, v in audio_sd.items() if k in model_dict and v.size() == model_dict[k].size()})
    not_loaded = [k for k in model_dict if k not in match]
    video_pos = ("pos_embed_spatial", "pos_embed_temporal")
    for k in video_pos + ("pos_embed_spatial_audio", "pos_embed_temporal_audio"):
        if k in model_dict and (k in video_sd or k in audio_sd):
            v = video_sd[k] if k in video_pos else audio_sd[k]
            t = model_dict[k].size()
            match[k] = F.interpolate(v.float().unsqueeze(0), (t[1], t[2]), mode="bilinear").squeeze(0)
            if k in not_loaded:
                not_loaded.remove(k)
    ms.load_state_dict(match, strict=False)
    if hasattr(ms, "_refresh_w16") and next(ms.parameters()).is_cuda:
        ms._refresh_w16()
    epoch = -1
    if "epoch" in video_ck and not epoch_reset:
        epoch = video_ck["epoch"]
        if optimizer is not None:
            optimizer.load_state_dict(video_ck["optimizer_state"])
        if scaler is not None and "scaler_state" in video_ck:
            scaler.load_state_dict(video_ck["scaler_state"])
    if report is not None:
        report.extend(not_loaded)
    return epoch


def load_train_checkpoint(cfg, model, optimizer, scaler=None):
    """checkpoint.py:617-659: auto-resume from OUTPUT_DIR, else TRAIN.CHECKPOINT_FILE_PATH (fine-tune init; with
    TRAIN.AUDIO_CHECKPOINT_FILE_PATH the video + audio pre-training pair of :645-656), else epoch 0."""
    if cfg.TRAIN.AUTO_RESUME and has_checkpoint(cfg.OUTPUT_DIR):
        epoch = load_checkpoint(get_last_checkpoint(cfg.OUTPUT_DIR), model, cfg.NUM_GPUS > 1, optimizer, scaler=scaler)
        return epoch + 1
    if cfg.TRAIN.CHECKPOINT_FILE_PATH != "" and getattr(cfg.TRAIN, "AUDIO_CHECKPOINT_FILE_PATH", "") != "":
        epoch = load_video_and_audio_checkpoints(cfg.TRAIN.CHECKPOINT_FILE_PATH, cfg.TRAIN.AUDIO_CHECKPOINT_FILE_PATH, model,
                                                 cfg.NUM_GPUS > 1, optimizer, scaler=scaler, epoch_reset=cfg.TRAIN.CHECKPOINT_EPOCH_RESET,
                                                 clear_name_pattern=getattr(cfg.TRAIN, "CHECKPOINT_CLEAR_NAME_PATTERN", ()))
        return epoch + 1
    if cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        epoch = load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, optimizer, scaler=scaler,
                                epoch_reset=cfg.TRAIN.CHECKPOINT_EPOCH_RESET,
                                clear_name_pattern=getattr(cfg.TRAIN, "CHECKPOINT_CLEAR_NAME_PATTERN", ()))
        return epoch + 1
    return 0
