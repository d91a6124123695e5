"""ctypes binding of libcsts_hip.so (C ABI declared in include/csts_hip.h).

The library is the product: if it is missing or fails to load, importing users get a loud
RuntimeError -- there is no CPU / eager fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_lib = None


class CstsError(RuntimeError):
    pass


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSTS_HIP_LIB") or os.path.join(_HERE, "libcsts_hip.so")   # override: A/B runs of two builds on one box
# The 16-bit activation type is a property of the LIBRARY BUILD (csts_amd/csrc/common.h): libcsts_hip.so = bfloat16,
# libcsts_hip_f16.so = IEEE half (the same sources with -DCSTS_HALF_F16).  One process runs ONE of them: set_half() picks it
# before the first load (CSTS_AMD.COMPUTE fp16 does, model.Runtime), and a later request for the other kind raises.
HALF = os.environ.get("CSTS_HALF", "bf16")
_PATHS = {"bf16": LIB_PATH, "fp16": os.environ.get("CSTS_HIP_LIB_F16") or os.path.join(_HERE, "libcsts_hip_f16.so")}


def set_half(kind: str):
    """Select the 16-bit type of this process ("bf16" | "fp16").  Must happen before the library is loaded with the other kind."""
    global HALF, LIB_PATH
    if kind not in _PATHS:
        raise ValueError(f"16-bit type must be 'bf16' or 'fp16', got {kind!r}")
    if _lib is not None and kind != HALF:
        raise CstsError(f"this process already runs the {HALF} kernel library; the {kind} build is a different shared object "
                        "(one 16-bit activation type per process: start a new process for the other mode)")
    HALF = kind
    LIB_PATH = _PATHS[kind]


def half_dtype():
    """torch dtype of the enum value BF16 (= CSTS_HALF, the 16-bit type of the loaded build)."""
    import torch
    return torch.float16 if HALF == "fp16" else torch.bfloat16


ABI_VERSION = 7   # == CSTS_ABI_VERSION of include/csts_hip.h this binding mirrors (struct layouts below)
F32, BF16 = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_DGELU = 0, 1, 2
MASK_NONE, MASK_SPATIAL = 0, 1

vp = C.c_void_p
i64 = C.c_int64
sz = C.c_size_t


class GemmArgs(C.Structure):
    _fields_ = [("layout", C.c_int),
                ("A", vp), ("a_dt", C.c_int), ("lda", i64),
                ("B", vp), ("b_dt", C.c_int), ("ldb", i64),
                ("C", vp), ("c_dt", C.c_int), ("ldc", i64),
                ("M", i64), ("N", i64), ("K", i64),
                ("bias", vp), ("epilogue", C.c_int),
                ("aux", vp), ("aux_dt", C.c_int), ("ldaux", i64),
                ("residual", vp), ("r_dt", C.c_int), ("ldr", i64), ("res_row_mod", i64),
                ("row_scale", vp), ("rows_per_scale", i64),
                ("compute", C.c_int), ("split_k", C.c_int),
                ("workspace", vp), ("ws_bytes", sz), ("colsum", vp), ("tile_rows", C.c_int), ("algo", C.c_int),
                ("res_up", C.c_int * 6)]


class DwconvGeom(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("HD", C.c_int),
                ("Tf", C.c_int), ("Hf", C.c_int), ("Wf", C.c_int),
                ("Tc", C.c_int), ("Hc", C.c_int), ("Wc", C.c_int),
                ("st", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("fine_batch_stride", i64), ("fine_token_stride", i64),
                ("coarse_batch_stride", i64), ("coarse_token_stride", i64)]


class DwconvWgradItem(C.Structure):
    _fields_ = [("geom", DwconvGeom), ("fine", vp), ("coarse", vp), ("workspace", vp)]


DWCONV_WGRAD_TABLE_ENTRY = 128


class PoolLnArgs(C.Structure):
    _fields_ = [("geom", DwconvGeom), ("nslots", C.c_int),
                ("fine", vp * 2), ("weight", vp * 2), ("gamma", vp * 2), ("beta", vp * 2),
                ("conv_out", vp * 2), ("y", vp * 2), ("mean", vp * 2), ("rstd", vp * 2),
                ("dt", C.c_int), ("eps", C.c_float)]


class PoolGeom(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int),
                ("Ti", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int),
                ("To", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("st", C.c_int), ("sh", C.c_int), ("sw", C.c_int)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("LSE", vp),
                ("dO", vp), ("delta", vp), ("dQ", vp), ("dK", vp), ("dV", vp),
                ("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("Nq", C.c_int), ("Nk", C.c_int),
                ("head_dim", C.c_int),
                ("q_strides", i64 * 3), ("k_strides", i64 * 3), ("v_strides", i64 * 3), ("o_strides", i64 * 3),
                ("do_strides", i64 * 3), ("dq_strides", i64 * 3), ("dk_strides", i64 * 3), ("dv_strides", i64 * 3),
                ("scale", C.c_float),
                ("mask_mode", C.c_int), ("mask_T", C.c_int), ("mask_HW", C.c_int)]


class Im2colGeom(C.Structure):
    _fields_ = [("B", C.c_int), ("Cin", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("kernel", C.c_int * 3), ("stride", C.c_int * 3), ("padding", C.c_int * 3),
                ("To", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("Kpad", C.c_int)]


class TokenSegment(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("src_bs", i64), ("dst_bs", i64), ("src_off", i64), ("dst_off", i64), ("n", C.c_int), ("pad_", C.c_int)]


class KvRowsGeom(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("Hc", C.c_int), ("Wc", C.c_int)]


class WgradItem(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("colsum", vp), ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("kbeg", i64), ("kend", i64), ("M", C.c_int), ("N", C.c_int), ("m0", C.c_int), ("n0", C.c_int)]


class ReduceDesc(C.Structure):
    _fields_ = [("ws", vp), ("out", vp), ("nrows", i64), ("ncols", i64), ("scale", C.c_float), ("pad_", C.c_int)]


class OptTensor(C.Structure):
    _fields_ = [("p", vp), ("m", vp), ("v", vp), ("w16", vp), ("n", i64), ("weight_decay", C.c_float), ("pad_", C.c_int)]


class OptArgs(C.Structure):
    _fields_ = [("chunk_tensor", vp), ("chunk_off", vp), ("nchunks", C.c_int), ("chunk_elems", C.c_int),
                ("tensors", vp), ("grads", vp), ("ntensors", C.c_int),
                ("partial", vp), ("state", vp), ("lr", vp),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("max_grad_norm", C.c_float),
                ("grad_dt", C.c_int), ("scaler", vp), ("growth", C.c_float), ("backoff", C.c_float), ("growth_interval", C.c_int),
                ("extra_sq", vp), ("n_extra_sq", C.c_int)]


class OptFactored(C.Structure):
    _fields_ = [("p", vp), ("m", vp), ("v", vp), ("w16", vp), ("dy", vp), ("a", vp), ("a_dt", C.c_int),
                ("N", C.c_int), ("K", C.c_int), ("T", C.c_int), ("weight_decay", C.c_float), ("pad_", C.c_int)]


# name -> (restype, argtypes); every symbol include/csts_hip.h declares
_I, _F = C.c_int, C.c_float
class TransposeTile(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("R", C.c_int), ("C", C.c_int), ("r0", C.c_int), ("c0", C.c_int)]


SYMBOLS = {
    "csts_last_error": (C.c_char_p, []),
    "csts_abi_version": (_I, []),
    "csts_half_kind": (_I, []),
    "csts_gemm": (_I, [C.POINTER(GemmArgs), vp]),
    "csts_gemm_splitk_workspace": (sz, [i64, i64, i64, _I]),
    "csts_gemm_v2_eligible": (_I, [C.POINTER(GemmArgs)]),
    "csts_wgrad_grouped": (_I, [vp, _I, _I, _I, vp]),
    "csts_wgrad_grouped8": (_I, [vp, _I, vp]),
    "csts_wgrad_grouped8_limited": (_I, [vp, _I, _I, vp]),
    "csts_wgrad_grouped5": (_I, [vp, _I, vp]),
    "csts_gemm_kernel_name": (_I, [C.POINTER(GemmArgs), C.c_char_p, _I, C.POINTER(_I)]),
    "csts_gemm_plan": (_I, [C.POINTER(GemmArgs), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "csts_layernorm_fwd": (_I, [vp, _I, vp, vp, vp, _I, vp, vp, i64, _I, _F, vp]),
    "csts_layernorm_fwd_add": (_I, [vp, vp, vp, _I, vp, vp, vp, _I, vp, vp, i64, _I, _F, vp]),
    "csts_layernorm_bwd_workspace": (sz, [i64, _I]),
    "csts_layernorm_bwd": (_I, [vp, _I, vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, vp, vp, sz, i64, _I, vp]),
    "csts_layernorm_bwd_ex": (_I, [vp, vp, _I, vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, i64, vp, vp, vp, sz, i64, _I, vp]),
    "csts_reduce_rows_batched": (_I, [vp, _I, i64, vp]),
    "csts_reduce_rows_wide": (_I, [vp, _I, i64, vp]),
    "csts_layernorm_bwd2": (_I, [vp, _I, vp, _I, vp, vp, vp, vp, vp, _I, vp, vp, vp, sz, i64, _I, vp]),
    "csts_reduce_rows": (_I, [vp, vp, i64, i64, _F, vp]),
    "csts_dwconv_strided": (_I, [C.POINTER(DwconvGeom), vp, _I, vp, vp, _I, vp]),
    "csts_dwconv_transposed": (_I, [C.POINTER(DwconvGeom), vp, _I, vp, vp, _I, vp]),
    "csts_dwconv_wgrad_workspace": (sz, [C.POINTER(DwconvGeom)]),
    "csts_dwconv_wgrad": (_I, [C.POINTER(DwconvGeom), vp, _I, vp, _I, vp, vp, sz, vp]),
    "csts_dwconv_transposed2": (_I, [C.POINTER(DwconvGeom), vp * 2, _I, vp * 2, vp * 2, _I, vp]),
    "csts_dwconv_wgrad2": (_I, [C.POINTER(DwconvGeom), vp * 2, _I, vp * 2, _I, vp * 2, vp, sz, vp]),
    "csts_dwconv_wgrad_grouped_workspace": (sz, [C.POINTER(DwconvGeom)]),
    "csts_dwconv_wgrad_grouped_plan": (_I, [C.POINTER(DwconvWgradItem), _I, vp, sz, C.POINTER(C.c_int)]),
    "csts_dwconv_wgrad_grouped": (_I, [vp, _I, _I, _I, vp]),
    "csts_pool_ln_fwd": (_I, [C.POINTER(PoolLnArgs), vp]),
    "csts_maxpool_fwd": (_I, [C.POINTER(PoolGeom), vp, _I, vp, vp, vp]),
    "csts_maxpool_bwd": (_I, [C.POINTER(PoolGeom), vp, _I, vp, vp, vp]),
    "csts_trilinear_fwd": (_I, [C.POINTER(PoolGeom), vp, _I, vp, _I, vp, _I, vp]),
    "csts_trilinear_bwd": (_I, [C.POINTER(PoolGeom), vp, _I, vp, _I, vp]),
    "csts_attn_fwd": (_I, [C.POINTER(AttnArgs), vp]),
    "csts_attn_bwd_workspace": (sz, [C.POINTER(AttnArgs)]),
    "csts_attn_bwd": (_I, [C.POINTER(AttnArgs), vp, sz, vp]),
    "csts_attn_probs": (_I, [C.POINTER(AttnArgs), vp, vp]),
    "csts_im2col": (_I, [C.POINTER(Im2colGeom), vp, _I, vp, _I, vp]),
    "csts_posembed_build": (_I, [vp, vp, vp, _I, _I, _I, vp]),
    "csts_transpose_batched": (_I, [vp, _I, vp, _I, i64, _I, _I, vp]),
    "csts_transpose_multi": (_I, [vp, _I, vp]),
    "csts_colsum_workspace": (sz, [i64, i64, i64]),
    "csts_colsum": (_I, [vp, _I, vp, vp, i64, i64, i64, vp, sz, vp]),
    "csts_axpby": (_I, [vp, _I, vp, _I, vp, _I, i64, _F, _F, vp]),
    "csts_rows_gather": (_I, [C.POINTER(KvRowsGeom), vp, _I, vp, vp]),
    "csts_copy_token_segments": (_I, [C.POINTER(TokenSegment), _I, _I, _I, _I, vp]),
    "csts_rows_scatter_add": (_I, [C.POINTER(KvRowsGeom), vp, _I, vp, _I, vp]),
    "csts_scale_rows": (_I, [vp, _I, vp, i64, vp, _I, i64, i64, vp]),
    "csts_add2": (_I, [vp, _I, vp, _I, vp, vp, i64, vp]),
    "csts_add2_scaled_copy": (_I, [vp, _I, vp, _I, vp, vp, vp, i64, i64, vp]),
    "csts_rowdot2": (_I, [vp, _I, vp, _I, vp, i64, _I, vp]),
    "csts_audio_attn_fwd": (_I, [vp, _I, vp, vp, _I, _I, _I, _I, _I, _F, vp]),
    "csts_audio_attn_bwd": (_I, [vp, _I, vp, vp, _I, _I, _I, _I, _I, _F, vp]),
    "csts_reweight_fwd": (_I, [vp, vp, vp, i64, _I, _I, vp]),
    "csts_reweight_bwd": (_I, [vp, vp, vp, vp, vp, i64, _I, _I, vp]),
    "csts_token_mean_fwd": (_I, [vp, vp, i64, _I, _I, vp]),
    "csts_token_mean_bwd": (_I, [vp, vp, i64, _I, _I, vp]),
    "csts_rowdot_fwd": (_I, [vp, _I, vp, vp, vp, i64, _I, vp]),
    "csts_rowdot_dx": (_I, [vp, vp, vp, _I, i64, _I, vp]),
    "csts_softmax_fwd": (_I, [vp, vp, i64, _I, _F, vp]),
    "csts_softmax_bwd": (_I, [vp, vp, vp, i64, _I, _F, vp]),
    "csts_kldiv_fwd": (_I, [vp, vp, vp, vp, i64, _I, _F, vp]),
    "csts_kldiv_bwd": (_I, [vp, vp, vp, vp, i64, _I, _F, vp]),
    "csts_rownorm_fwd": (_I, [vp, vp, vp, i64, _I, _F, vp]),
    "csts_rownorm_bwd": (_I, [vp, vp, vp, vp, i64, _I, _F, vp]),
    "csts_egonce_fwd": (_I, [vp, vp, vp, vp, _I, _F, vp]),
    "csts_egonce_bwd": (_I, [vp, vp, vp, vp, vp, _I, _F, vp]),
    "csts_adamw_step": (_I, [C.POINTER(OptArgs), vp]),
    "csts_factored_sqnorm_workspace": (sz, [C.POINTER(OptFactored), _I]),
    "csts_factored_sqnorm": (_I, [C.POINTER(OptFactored), _I, vp, vp, sz, vp]),
    "csts_adamw_factored": (_I, [C.POINTER(OptFactored), _I, vp, vp, _F, _F, _F, vp]),
    "csts_frames_normalize": (_I, [vp, vp, _I, i64, _I, C.c_float * 3, C.c_float * 3, vp]),
    "csts_stft_frames": (_I, [_I, _I, _I]),
    "csts_stft_logpower": (_I, [vp, vp, _I, _I, _I, _I, _I, _F, vp]),
    "csts_audio_windows": (_I, [vp, vp, vp, _I, _I, _I, _I, _I, vp]),
    "csts_gaze_heatmaps": (_I, [vp, _I, vp, i64, _I, _I, _I, vp]),
    "csts_adaptive_f1_workspace": (sz, [i64, _I]),
    "csts_adaptive_f1": (_I, [vp, vp, vp, vp, _I, i64, _I, _I, vp, vp, sz, vp]),
}



def load() -> C.CDLL:
    """Load libcsts_hip.so (once).  Raises loudly when it is absent: build it with
    ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C csts_amd/csrc``."""
    global _lib
    if _lib is not None:
        return _lib
    if HALF not in _PATHS:
        raise CstsError(f"CSTS_HALF must be 'bf16' or 'fp16', got {HALF!r}")
    if not os.path.exists(LIB_PATH):
        raise CstsError(f"{LIB_PATH} not found: the HIP kernel library is required (no CPU fallback). "
                        "Build it with `make -C csts_amd/csrc` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise CstsError(f"{LIB_PATH} does not export {name}: stale build?")
        fn.restype = res
        fn.argtypes = args
    if lib.csts_half_kind() != (1 if HALF == "fp16" else 0):
        raise CstsError(f"{LIB_PATH} holds the {'fp16' if lib.csts_half_kind() else 'bf16'} kernels but this process selected {HALF}")
    got = lib.csts_abi_version()
    if got != ABI_VERSION:
        # a stale / alternate build reads the argument structs with another layout (e.g. a pre-res_up library would read
        # the coarse decoder skip as a plain residual: out-of-bounds reads, silently wrong outputs) -- never run it
        raise CstsError(f"{LIB_PATH} has C-ABI version {got}, this binding needs {ABI_VERSION}: rebuild it "
                        "(`make -C csts_amd/csrc`) or point CSTS_HIP_LIB at a matching build")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().csts_last_error()
        raise CstsError(f"{what} failed: {msg.decode() if msg else rc}")
