"""On-device input pipeline: the steps the reference runs on the CPU right before the model --
frame normalisation (slowfast/datasets/utils.py:290-307), the log-power STFT of the 24 kHz audio (data/preprocess.py:276-290,
offline with librosa there), the T spectrogram windows around the sampled frames (ego4d_avgaze_forecast.py:214-219) and
the Gaussian gaze heat maps (ego4d_avgaze_forecast.py:318-326,404-422) -- as HIP kernels, so a batch can be assembled
from uint8 frames, a waveform and gaze points without leaving the GPU (the fp32 clip never crosses PCIe)."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _s():
    return torch.cuda.current_stream().cuda_stream


def _gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise L.CstsError("csts_amd.inputs runs on MI355X only: inputs must be GPU tensors")


def normalize_frames(frames_u8: torch.Tensor, mean=(0.45, 0.45, 0.45), std=(0.225, 0.225, 0.225)) -> torch.Tensor:
    """uint8 (B, T, H, W, C) -> fp32 (B, C, T, H, W) = (x/255 - mean)/std."""
    _gpu(frames_u8)
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 5
    B, T, H, W, Cc = frames_u8.shape
    x = frames_u8.contiguous()
    out = torch.empty(B, Cc, T, H, W, dtype=torch.float32, device=x.device)
    f3 = C.c_float * 3
    L.check(L.load().csts_frames_normalize(x.data_ptr(), out.data_ptr(), B, T * H * W, Cc, f3(*mean), f3(*std), _s()),
            "csts_frames_normalize")
    return out


def stft_logpower(wav: torch.Tensor, n_fft: int = 511, hop: int = 120, win: int = 240, eps: float = 1e-6) -> torch.Tensor:
    """fp32 waveform (B, n) -> log(|STFT|^2 + eps), (B, n_fft//2 + 1, frames) with librosa.stft semantics."""
    _gpu(wav)
    w = wav.contiguous().float()
    B, n = w.shape
    lib = L.load()
    nfr = lib.csts_stft_frames(n, n_fft, hop)
    spec = torch.empty(B, n_fft // 2 + 1, nfr, dtype=torch.float32, device=w.device)
    L.check(lib.csts_stft_logpower(w.data_ptr(), spec.data_ptr(), B, n, n_fft, hop, win, eps, _s()), "csts_stft_logpower")
    return spec


def audio_windows(spec: torch.Tensor, frames_idx: torch.Tensor, frame_length: float, width: int = 256) -> torch.Tensor:
    """(B, nbins, cols) spectrogram + per-clip frame indices (B, T) -> (B, 1, T, nbins, width) windows centred at
    round(idx / frame_length * cols), clipped to the valid range (ego4d_avgaze_forecast.py:216-219)."""
    _gpu(spec, frames_idx)
    B, nbins, cols = spec.shape
    T = frames_idx.shape[1]
    centers = torch.round(frames_idx.double() / frame_length * cols).to(torch.int32).clamp_(width // 2, cols - 1 - width // 2)
    centers = centers.contiguous()
    out = torch.empty(B, 1, T, nbins, width, dtype=torch.float32, device=spec.device)
    L.check(L.load().csts_audio_windows(spec.contiguous().data_ptr(), centers.data_ptr(), out.data_ptr(), B, T, nbins, cols, width,
                                        _s()), "csts_audio_windows")
    return out


def gaze_heatmaps(labels: torch.Tensor, H: int = 64, W: int = 64, ksize: int = 19) -> torch.Tensor:
    """labels (B, T, >=2) with x, y in [0, 1] -> (B, T, H, W) maps summing to 1 per frame."""
    _gpu(labels)
    lab = labels.contiguous().float()
    B, T, S = lab.shape
    out = torch.empty(B, T, H, W, dtype=torch.float32, device=lab.device)
    L.check(L.load().csts_gaze_heatmaps(lab.data_ptr(), S, out.data_ptr(), B * T, H, W, ksize, _s()), "csts_gaze_heatmaps")
    return out


def assemble_batch(frames_u8, wav, frames_idx, frame_length, labels):
    """uint8 frames (B, T, H, W, 3), waveform (B, n), sampled frame indices (B, T), gaze labels (B, T, 3) -> the batch
    dict the training step takes (video, audio, labels_hm, labels)."""
    return {"video": normalize_frames(frames_u8),
            "audio": audio_windows(stft_logpower(wav), frames_idx, frame_length),
            "labels_hm": gaze_heatmaps(labels), "labels": labels}
