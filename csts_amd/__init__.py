"""csts_amd: MI355X-native (gfx950) implementation of the CSTS audio-visual gaze-anticipation training path.

Python host code on PyTorch-ROCm over a C-ABI kernel library (csts_amd/libcsts_hip.so, include/csts_hip.h).
Importing the package is cheap and GPU-free; the kernel library is loaded on first use and its absence is a
hard error (there is no CPU fallback)."""
from .registry import MODEL_REGISTRY  # noqa: F401
from .config import get_cfg, load_yaml, assert_and_infer_cfg  # noqa: F401
from .build import build_model  # noqa: F401

__all__ = ["MODEL_REGISTRY", "build_model", "get_cfg", "load_yaml", "assert_and_infer_cfg"]
