"""ctypes binding of libunitspeech_hip.so (the C ABI of include/unitspeech_hip.h).

There is deliberately no fallback: if the library is missing it is built with hipcc, and if that fails
(or a call returns an error code) a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _build

US_OK = 0
US_CREATE_EXACT_FP32 = 1
US_RANGE_ACT, US_RANGE_WEIGHT = 1, 2
US_BACKWARD_GRADS_ZEROED, US_BACKWARD_KEEP_TAPE = 1, 2
ERRORS = {-1: "EINVAL", -2: "ENOKEY", -3: "ESHAPE", -4: "EWEIGHTS", -5: "EWORKSPACE", -6: "EHIP"}


class us_config(C.Structure):
    _fields_ = [("n_feats", C.c_int32), ("dim", C.c_int32), ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 6),
                ("spk_emb_dim", C.c_int32), ("beta_min", C.c_float), ("beta_max", C.c_float), ("pe_scale", C.c_float)]


class us_encoder_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_vocab", "n_feats", "n_channels", "filter_channels", "n_heads", "n_layers", "kernel_size",
                                         "window_size")]


class us_duration_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_channels", "filter_channels", "kernel_size", "spk_emb_dim")]


# symbol -> (restype, argtypes); must list every function declared in include/unitspeech_hip.h
SIGNATURES = {
    "us_decoder_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(us_config)]),
    "us_decoder_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(us_config), C.c_uint]),
    "us_decoder_destroy": (C.c_int, [C.c_void_p]),
    "us_range_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.c_int, C.c_void_p]),
    "us_range_status_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "us_decoder_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    "us_decoder_flush_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "us_decoder_set_training": (C.c_int, [C.c_void_p, C.c_int]),
    "us_decoder_stale_inference_forms": (C.c_int, [C.c_void_p]),
    "us_decoder_num_weights": (C.c_int, [C.c_void_p]),
    "us_decoder_num_loaded": (C.c_int, [C.c_void_p]),
    "us_decoder_weight_key": (C.c_char_p, [C.c_void_p, C.c_int]),
    "us_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "us_sampler_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "us_estimator_forward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_reverse_diffusion": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                                          C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p,
                                                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_step_coefficients": (C.c_int, [C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "us_fill_normal": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p]),
    "us_estimator_flops": (C.c_double, [C.c_void_p, C.c_int]),
    "us_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "us_estimator_forward_train": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                                               C.POINTER(C.c_uint64), C.c_void_p]),
    "us_estimator_backward": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "us_tape_release": (C.c_int, [C.c_void_p, C.c_uint64]),
    "us_grad_is_overwritten": (C.c_int, [C.c_void_p, C.c_char_p]),
    "us_forward_diffusion": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "us_diffusion_loss_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "us_diffusion_loss": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_pow2_scale": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "us_mul_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "us_finetune_segment": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_void_p]),
    "us_tts_durations": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "us_tts_align": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]),
    "us_encoder_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(us_encoder_config)]),
    "us_duration_predictor_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(us_duration_config)]),
    "us_frontend_destroy": (C.c_int, [C.c_void_p]),
    "us_frontend_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    "us_frontend_num_weights": (C.c_int, [C.c_void_p]),
    "us_frontend_weight_key": (C.c_char_p, [C.c_void_p, C.c_int]),
    "us_frontend_last_error": (C.c_char_p, [C.c_void_p]),
    "us_frontend_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "us_encoder_forward": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_duration_predictor_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_debug_block": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "us_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "us_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "us_profile_read_f16": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "us_clip_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "us_last_error": (C.c_char_p, [C.c_void_p]),
}

_lock = threading.Lock()
_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """dlopen the in-tree library, building it first when absent or older than any of its sources (a cheap mtime check; with
    no hipcc on the machine an existing library is used as it is)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB
        alt = os.environ.get("UNITSPEECH_AMD_LIB")          # experiments only: another build of the same ABI (A/B runs on one GPU box)
        if alt:
            if not os.path.exists(alt):
                raise RuntimeError(f"UNITSPEECH_AMD_LIB={alt} does not exist")
            path, build_if_missing = alt, False
        if build_if_missing:
            try:
                path = _build.build_library(force=os.environ.get("UNITSPEECH_AMD_REBUILD") == "1")
            except RuntimeError:
                if not os.path.exists(path) or _build.have_hipcc():
                    raise
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found and could not be built: the HIP decoder has no CPU fallback")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here == missing export
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int, handle=None, what: str = "") -> None:
    if rc == US_OK:
        return
    lib = load()
    msg = lib.us_last_error(handle)
    raise RuntimeError(f"libunitspeech_hip: {what} failed with {ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")
