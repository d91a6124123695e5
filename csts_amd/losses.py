"""Loss modules with the reference's names and call signatures (slowfast/models/losses.py:51-82,152-170,187-207;
slowfast/utils/utils.py:5-24), computed by the HIP kernels."""
import torch.nn as nn

from . import ops


def frame_softmax(logits, temperature):
    """slowfast/utils/utils.py:5-12."""
    return ops.frame_softmax(logits, temperature)


def sim_matrix(a, b, eps=1e-8):
    """slowfast/utils/utils.py:15-24."""
    return ops.sim_matrix(a, b, eps)


class KLDiv(nn.Module):
    """KL divergence for 3D attention maps (losses.py:51-82)."""

    def forward(self, pred, target=None):
        return ops.kldiv(pred, target)


class EgoNCE(nn.Module):
    """Symmetric InfoNCE over a similarity matrix (losses.py:152-170).  Works on any device the kernels run
    on -- the reference's hard-coded ``torch.eye(...).cuda()`` (:158) has no counterpart here."""

    def __init__(self, temperature=0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, x):
        return ops.egonce(x, self.temperature)


_LOSSES = {"kldiv": KLDiv, "egonce": EgoNCE}


def get_loss_func(loss_name):
    """losses.py:199-207 (only the losses on the CSTS path are provided)."""
    if loss_name not in _LOSSES:
        raise NotImplementedError("Loss {} is not supported".format(loss_name))
    return _LOSSES[loss_name]
