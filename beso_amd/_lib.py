"""ctypes binding of libbeso_hip.so (include/beso_hip.h).  The product path has no CPU fallback:
if the library is missing or fails to load, importing callers get a RuntimeError."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# BESO_HIP_LIB points the binding at another build of the same library (kernel experiments: tools/variants.py)
LIB_PATH = os.environ.get("BESO_HIP_LIB") or os.path.join(_HERE, "lib", "libbeso_hip.so")

PREC_BF16, PREC_FP32, PREC_BF16X3, PREC_FP16 = 0, 1, 2, 3
PRECISIONS = {"bf16": PREC_BF16, "fp32": PREC_FP32, "bf16x3": PREC_BF16X3, "fp16": PREC_FP16}
FLAG_UNCOND = 1
# execution-plan hints of the forward calls (include/beso_hip.h: which kernels run, never what they compute)
PLAN_PER_OP, PLAN_BLOCKS, PLAN_SMALL, PLAN_FUSED = 0x10, 0x20, 0x40, 0x80
PLAN_SPW2, PLAN_SPW4, PLAN_SPW8 = 0x100, 0x200, 0x300
SAMPLE_STEPWISE = 0x1000
TRAIN_LAST_ACTION_ONLY, TRAIN_PLAN_PER_OP, TRAIN_PLAN_TILES = 1, 2, 4
SAMPLER_IDS = {"ddim": 0, "euler": 1, "heun": 2}
GOAL_RANDOM, GOAL_TAIL, GOAL_SEQ_END = 0, 1, 2
STEP_DDIM, STEP_EULER, STEP_HEUN_PREDICT, STEP_HEUN_CORRECT = 0, 1, 2, 3
SITES = {"off": 0, "gemm_qkv": 1, "gemm_proj": 2, "gemm_fc1": 3, "gemm_fc2": 4, "attention": 5,
         "layernorm": 6, "embed": 7, "head": 8, "forward": 9, "fused_layer": 10, "small": 11}

# every symbol include/beso_hip.h declares (tests check that the library exports all of them)
EXPORTS = ["beso_version", "beso_status_string", "beso_last_error", "beso_num_params", "beso_packed_bytes", "beso_pack_weights",
           "beso_workspace_bytes", "beso_score_fwd", "beso_denoise_fwd", "beso_sampler_step", "beso_sample",
           "beso_profile_enable", "beso_profile_read", "beso_adam_step",
           "beso_train_workspace_bytes", "beso_grad_floats", "beso_loss_grad", "beso_gather_windows",
           "beso_loss_grad_overlap", "beso_grad_early_range", "beso_sample_ancestral", "beso_goal_mask",
           "beso_loss_grad_streams", "beso_log_logistic", "beso_scale_rows"]
# include/beso_hip_debug.h: the development build only (libbeso_hip_dev.so); the product library exports none of them
DEV_EXPORTS = ["beso_debug_set_stamps", "beso_debug_gemm"]
DEV_LIB_PATH = os.path.join(_HERE, "lib", "libbeso_hip_dev.so")


class BesoConfig(C.Structure):
    """struct beso_config (include/beso_hip.h)."""
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("embed_dim", C.c_int32),
                ("n_layers", C.c_int32), ("n_heads", C.c_int32), ("goal_seq_len", C.c_int32),
                ("obs_seq_len", C.c_int32), ("linear_output", C.c_int32), ("sigma_data", C.c_float)]


class BesoHipError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load the HIP library (once).  Fails loudly -- there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process BEFORE this
        # library is mapped, so that both resolve to the same runtime instance (streams and device
        # pointers are handed across).  Loading ours first would pull in /opt/rocm's copy as a second
        # runtime and every launch on a torch stream would fail.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise BesoHipError(
                f"{LIB_PATH} not found: build it with `python -m beso_amd.build` (hipcc, gfx950). "
                "beso_amd has no CPU or eager fallback for the score-denoising path.")
        lib = C.CDLL(LIB_PATH)
        vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        cfgp = C.POINTER(BesoConfig)
        lib.beso_version.restype = C.c_char_p
        lib.beso_version.argtypes = []
        lib.beso_status_string.restype = C.c_char_p
        lib.beso_status_string.argtypes = [i32]
        lib.beso_last_error.restype = C.c_char_p
        lib.beso_last_error.argtypes = []
        lib.beso_num_params.restype = i32
        lib.beso_num_params.argtypes = [cfgp]
        lib.beso_packed_bytes.restype = sz
        lib.beso_packed_bytes.argtypes = [cfgp, i32]
        lib.beso_pack_weights.restype = i32
        lib.beso_pack_weights.argtypes = [cfgp, C.POINTER(vp), i32, vp, sz, i32, vp]
        lib.beso_workspace_bytes.restype = sz
        lib.beso_workspace_bytes.argtypes = [cfgp, i32, i32, i32, i32]
        lib.beso_score_fwd.restype = i32
        lib.beso_score_fwd.argtypes = [cfgp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]
        lib.beso_denoise_fwd.restype = i32
        lib.beso_denoise_fwd.argtypes = [cfgp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, sz, vp]
        lib.beso_sampler_step.restype = i32
        lib.beso_sampler_step.argtypes = [i32, vp, vp, vp, vp, vp, f32, f32, sz, vp]
        lib.beso_sample.restype = i32
        lib.beso_sample.argtypes = [cfgp, vp, i32, i32, vp, vp, vp, i32, i32, C.POINTER(C.c_float), i32, f32, i32,
                                    vp, sz, vp]
        if hasattr(lib, "beso_sample_ancestral") or not os.environ.get("BESO_HIP_LIB"):
            lib.beso_sample_ancestral.restype = i32
            lib.beso_sample_ancestral.argtypes = [cfgp, vp, i32, vp, vp, vp, i32, i32, C.POINTER(C.c_float), i32, f32, f32, vp,
                                                  i32, vp, sz, vp]
        lib.beso_profile_enable.restype = None
        lib.beso_profile_enable.argtypes = [i32]
        if hasattr(lib, "beso_adam_step") or not os.environ.get("BESO_HIP_LIB"):   # (A/B builds of older revisions)
            lib.beso_adam_step.restype = i32
            lib.beso_adam_step.argtypes = [vp, i32, vp, vp, vp, f32, f32, f32, f32, f32, i32, i32, f32, vp]
        if hasattr(lib, "beso_loss_grad") or not os.environ.get("BESO_HIP_LIB"):
            lib.beso_train_workspace_bytes.restype = sz
            lib.beso_train_workspace_bytes.argtypes = [cfgp, i32, i32, i32]
            lib.beso_grad_floats.restype = sz
            lib.beso_grad_floats.argtypes = [cfgp]
            lib.beso_loss_grad.restype = i32
            lib.beso_loss_grad.argtypes = [cfgp, C.POINTER(vp), i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32,
                                           f32, f32, C.c_uint, f32, vp, sz, vp]
            lib.beso_goal_mask.restype = i32
            lib.beso_goal_mask.argtypes = [vp, i32, i32, i32, f32, C.c_uint, vp]
            if hasattr(lib, "beso_loss_grad_overlap") or not os.environ.get("BESO_HIP_LIB"):
                lib.beso_loss_grad_overlap.restype = i32
                lib.beso_loss_grad_overlap.argtypes = lib.beso_loss_grad.argtypes + [vp]
                if hasattr(lib, "beso_loss_grad_streams") or not os.environ.get("BESO_HIP_LIB"):     # (HipTrainStep.run probes it too)
                    lib.beso_loss_grad_streams.restype = i32
                    lib.beso_loss_grad_streams.argtypes = lib.beso_loss_grad.argtypes + [vp, vp]
                lib.beso_grad_early_range.restype = i32
                lib.beso_grad_early_range.argtypes = [cfgp, C.POINTER(sz), C.POINTER(sz)]
        if hasattr(lib, "beso_scale_rows") or not os.environ.get("BESO_HIP_LIB"):
            lib.beso_scale_rows.restype = i32
            lib.beso_scale_rows.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
        if hasattr(lib, "beso_log_logistic") or not os.environ.get("BESO_HIP_LIB"):
            lib.beso_log_logistic.restype = i32
            lib.beso_log_logistic.argtypes = [vp, vp, sz, C.c_double, C.c_double, C.c_double, C.c_double, vp]
        if hasattr(lib, "beso_gather_windows") or not os.environ.get("BESO_HIP_LIB"):
            lib.beso_gather_windows.restype = i32
            lib.beso_gather_windows.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, C.c_longlong, vp, vp, i32, i32, i32,
                                                i32, i32, vp, vp, vp, vp]
        lib.beso_profile_read.restype = i32
        lib.beso_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i32)]
        _lib = lib
    return _lib


_dev = None


def load_dev() -> C.CDLL:
    """The development build (`python -m beso_amd.build --dev`, include/beso_hip_debug.h): phase stamps and the GEMM layout
    probe.  A separate library image with its own state; nothing in the package uses it."""
    global _dev
    if _dev is None:
        import torch  # noqa: F401      (the HIP runtime of the process: see load())
        if not os.path.exists(DEV_LIB_PATH):
            raise BesoHipError(f"{DEV_LIB_PATH} not found: build it with `python -m beso_amd.build --dev`")
        lib = C.CDLL(DEV_LIB_PATH)
        lib.beso_debug_set_stamps.restype = None
        lib.beso_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
        lib.beso_debug_gemm.restype = C.c_int
        lib.beso_debug_gemm.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
        _dev = lib
    return _dev


def check(status: int, what: str = "") -> None:
    """Non-zero status -> exception.  Shape/config/argument errors are ValueError, matching the
    reference's conventions (ValueError for unknown sampler / schedule: beso_agent.py:455,598;
    `assert t <= block_size`: score_gpts.py:282); runtime failures are BesoHipError."""
    if status == 0:
        return
    msg = load().beso_status_string(status).decode()
    text = f"beso_hip: {what + ': ' if what else ''}{msg} (status {status})"
    if status in (-1, -2, -3, -5):
        raise ValueError(text)
    if status == -6:
        text += ": " + load().beso_last_error().decode()
    raise BesoHipError(text)
