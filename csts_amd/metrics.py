"""Evaluation metric of the gaze heat maps on the device: ``adaptive_f1`` with the reference's name, arguments and return
value (slowfast/utils/metrics.py:9-74), computed by csts_adaptive_f1 without the (n_thresholds, B, T, H, W) temporaries
of the reference.  ``rescale=True`` folds in the per-frame min-max rescale its callers apply first
(tools/test_avgaze_net.py:66-68, tools/train_avgaze_net.py:125-127)."""
from __future__ import annotations

import numpy as np
import torch

from . import lib as L

_FIXATION_0 = ("ego4dgaze", "ego4dgaze_forecast", "ego4d_av_gaze", "ego4d_av_gaze_forecast", "aria_gaze",
               "aria_gaze_forecast", "aria_av_gaze", "aria_av_gaze_forecast")


def thresholds_for(dataset: str) -> np.ndarray:
    """metrics.py:35-43: the search space depends on the dataset."""
    if "forecast" in dataset and "aria" not in dataset:
        return np.linspace(0.01, 0.07, 31)
    if "forecast" in dataset and "aria" in dataset:
        return np.linspace(0.0, 0.02, 21)
    return np.linspace(0, 0.02, 11)


def adaptive_f1(preds, labels_hm, labels, dataset, rescale: bool = False):
    """preds (B, 1, T, H, W) heat maps (already min-max rescaled unless rescale=True), labels_hm (B, T, H, W),
    labels (B, T, 3) with the gaze type in [..., 2].  Returns (f1, recall, precision, threshold) as Python floats."""
    if not preds.is_cuda:
        raise L.CstsError("csts_amd.metrics.adaptive_f1 runs on MI355X only: inputs must be GPU tensors")
    if dataset == "egteagaze":
        fixation_idx = 1
    elif dataset in _FIXATION_0:
        fixation_idx = 0
    else:
        raise NotImplementedError(f"Metrics of {dataset} is not implemented.")
    thr = thresholds_for(dataset)
    dev = preds.device
    p = preds.detach().squeeze(1).contiguous().float()
    q = labels_hm.detach().contiguous().float()
    B, T, H, W = q.shape
    assert p.shape == q.shape, (p.shape, q.shape)
    tracked = (labels.detach().reshape(B * T, -1)[:, 2] == fixation_idx).to(torch.uint8).contiguous()
    thr_d = torch.tensor(thr.astype(np.float32), device=dev)      # torch compares an fp32 tensor with the scalar in fp32
    out = torch.empty(4, dtype=torch.float32, device=dev)
    lib = L.load()
    ws = torch.empty(max(int(lib.csts_adaptive_f1_workspace(B * T, len(thr))), 16), dtype=torch.uint8, device=dev)
    L.check(lib.csts_adaptive_f1(p.data_ptr(), q.data_ptr(), tracked.data_ptr(), thr_d.data_ptr(), len(thr), B * T, H * W,
                                 1 if rescale else 0, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream), "csts_adaptive_f1")
    f1, rec, prec, idx = out.cpu().tolist()
    return float(f1), float(rec), float(prec), thr[int(idx)]
