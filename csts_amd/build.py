"""build_model: the drop-in boundary (slowfast/models/build.py:18-47).

Same contract: look the class up by cfg.MODEL.MODEL_NAME, construct it with cfg, move it to the current
GPU, and for NUM_GPUS > 1 wrap it for data-parallel training.  The wrapper is this package's own
bucketed RCCL gradient all-reduce (csts_amd.distributed.GradAllReduce) instead of torch DDP; like DDP it
exposes the wrapped model as ``.module``."""
import torch

from .registry import MODEL_REGISTRY
from . import model as _model  # noqa: F401  (registers CSTS)


def build_model(cfg, gpu_id=None):
    if torch.cuda.is_available():
        assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    else:
        assert cfg.NUM_GPUS == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if cfg.NUM_GPUS:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)
    if cfg.NUM_GPUS > 1:
        from .distributed import GradAllReduce
        model = GradAllReduce(model, bucket_mb=cfg.CSTS_AMD.GRAD_BUCKET_MB)
    return model
