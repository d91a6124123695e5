from csts_amd.train import construct_optimizer, set_lr  # noqa: F401
from csts_amd.train import get_lr_at_epoch as _lr


def get_epoch_lr(cur_epoch, cfg):
    return _lr(cfg, cur_epoch)
