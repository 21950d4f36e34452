"""ctypes binding of libsntc_hip.so (include/sntc.h).

There is no CPU fallback: importing this module without the built library raises, and every call
that returns a non-zero status raises :class:`SntcError` carrying ``sntc_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SNTC_LIB", _PKG / "lib" / "libsntc_hip.so"))

OK, ERR_BAD_SHAPE, ERR_UNSUPPORTED, ERR_HIP, ERR_NONFINITE, ERR_NO_DEVICE = range(6)

CONV2D, CONV2D_TRANSPOSE, SIGNAL_DOWN, SIGNAL_UP = range(4)
ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_SIGMOID = range(4)
PRO_NONE, PRO_ABS, PRO_SQUARE = range(3)
EPI_STORE, EPI_ADD, EPI_GATE, EPI_RES_DIV, EPI_RES_MUL, EPI_RES_DIV_SQRT, EPI_RES_MUL_SQRT, EPI_MASK_RELU, EPI_MASK_LEAKY = range(9)


class SntcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sntc error {code}: {msg}")
        self.code = code


class NonFiniteError(SntcError, ValueError):
    """Mirrors tf.debugging.check_numerics' InvalidArgumentError (reference mshyper/models.py:308-309,356)."""


class ConvDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("act", C.c_int32), ("prologue", C.c_int32),
                ("epilogue", C.c_int32), ("reserved", C.c_int32 * 7)]


class SynBatch(C.Structure):
    """sntc_syn_batch: one group of same-sized images of a sntc_syn_forward call."""
    _fields_ = [("y_hat", C.c_void_p), ("hidden", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("reserved", C.c_int32)]


# name -> (restype, argtypes); every symbol include/sntc.h declares
_P = C.c_void_p
SIGNATURES = {
    "sntc_last_error": (C.c_char_p, []),
    "sntc_version": (C.c_int, []),
    "sntc_device_count": (C.c_int, []),
    "sntc_device_arch": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "sntc_conv_plan_create": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, C.POINTER(_P)]),
    "sntc_conv_plan_destroy": (None, [_P]),
    "sntc_conv_plan_update": (C.c_int, [_P, _P, _P, _P]),
    "sntc_plan_group_create": (C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    "sntc_plan_group_update": (C.c_int, [_P, _P]),
    "sntc_plan_group_destroy": (None, [_P]),
    "sntc_conv_out_shape": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sntc_conv_flops": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_conv_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "sntc_conv_workspace_bytes": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_conv_fusable": (C.c_int, [_P, _P]),
    "sntc_conv_fused_workspace_bytes": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_conv_forward_fused": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "sntc_resblock_supported": (C.c_int, [C.c_int]),
    "sntc_resblock_plan_create": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P, C.POINTER(_P)]),
    "sntc_resblock_plan_update": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sntc_resblock_plan_destroy": (None, [_P]),
    "sntc_resblock_flops": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_resblock_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_resblock_plan_set_workgroups": (C.c_int, [_P, C.c_int]),
    "sntc_upsmall_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sntc_upsmall_plan_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.POINTER(_P)]),
    "sntc_upsmall_plan_update": (C.c_int, [_P, _P, _P, _P]),
    "sntc_upsmall_plan_destroy": (None, [_P]),
    "sntc_upsmall_flops": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_upsmall_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_rgbconv_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sntc_rgbconv_plan_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.POINTER(_P)]),
    "sntc_rgbconv_plan_update": (C.c_int, [_P, _P, _P, _P]),
    "sntc_rgbconv_plan_destroy": (None, [_P]),
    "sntc_rgbconv_flops": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_rgbconv_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_rgbconv_plan_set_workgroups": (C.c_int, [_P, C.c_int]),
    "sntc_syn_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sntc_syn_plan_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(_P)]),
    "sntc_syn_plan_update": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "sntc_syn_plan_destroy": (None, [_P]),
    "sntc_syn_flops": (C.c_int64, [_P, C.c_int64]),
    "sntc_syn_workspace_bytes": (C.c_int64, [_P]),
    "sntc_syn_forward": (C.c_int, [_P, _P, C.c_int, _P, C.c_size_t, _P]),
    "sntc_syn_plan_set_workgroups": (C.c_int, [_P, C.c_int]),
    "sntc_syn_plan_units": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int]),
    "sntc_syn_describe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "sntc_syn_selfcheck": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_double)]),
    "sntc_conv_plan_set_tile": (C.c_int, [C.c_void_p, C.c_int]),
    "sntc_conv_plan_set_schedule": (C.c_int, [C.c_void_p, C.c_int]),
    "sntc_split3": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P]),
    "sntc_dequant_split3": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P, _P]),
    "sntc_conv_status": (C.c_int, [C.POINTER(C.c_int), _P]),
    "sntc_conv_status_inject": (C.c_int, [C.c_int, _P]),
    "sntc_conv_set_stream_k": (C.c_int, [C.c_int]),
    "sntc_conv_get_stream_k": (C.c_int, []),
    "sntc_conv_launch_info": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sntc_conv_launch_order": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "sntc_conv_tune_workspace_bytes": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    "sntc_conv_plan_tune": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), _P]),
    "sntc_conv_plan_clear_tuning": (C.c_int, [_P]),
    "sntc_conv_plan_candidates": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "sntc_conv_plan_set_choice": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sntc_gdn_small": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_two_layer_tail": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P,
                                      C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_two_layer_tail_pixels": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "sntc_pad_reflect": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_pad_zero": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_crop": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_depth_to_space": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_concat_channels": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int64, _P, _P]),
    "sntc_pixels_sse": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "sntc_float_sse": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_prior_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), C.POINTER(C.c_float), _P, C.POINTER(_P)]),
    "sntc_prior_destroy": (None, [_P]),
    "sntc_entropy_factorized": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.c_int, _P]),
    "sntc_entropy_scale_normal": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, C.c_int, _P]),
    "sntc_dequant_scale_normal": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "sntc_rans_cap_words": (C.c_int64, [C.c_int64, C.c_int]),
    "sntc_rans_encode": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int64, _P, _P, _P]),
    "sntc_rans_compact": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int, _P, _P]),
    "sntc_rans_lut_budget": (C.c_int64, [C.c_int, C.c_int]),
    "sntc_rans_decode": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P]),
    "sntc_scale_table_ids": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P]),
    "sntc_channel_table_ids": (C.c_int, [C.c_int64, C.c_int, _P, _P]),
    "sntc_round_to_int": (C.c_int, [_P, C.c_int64, _P, _P]),
    "sntc_int_to_float": (C.c_int, [_P, C.c_int64, _P, _P]),
    "sntc_ssim_scale": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
    "sntc_avgpool2_symmetric": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_pixels_float": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_sga_factorized_fwd": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_float, _P, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P]),
    "sntc_sga_normal_fwd": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, C.c_float, _P, C.c_uint64, C.c_uint64, _P, _P, _P,
                                      _P, _P, _P]),
    "sntc_sga_normal_bwd": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int64, C.c_int, _P, _P, _P]),
    "sntc_uq_sample": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "sntc_sga_chain": (C.c_int, [_P, _P, _P, C.c_float, C.c_int64, _P, _P]),
    "sntc_distortion_grad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
    "sntc_two_layer_tail_bwd": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
    "sntc_two_layer_out_adjoint": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_float, _P]),
    "sntc_conv_wgrad_workspace_bytes": (C.c_int64, [C.c_int] * 9),
    "sntc_conv_wgrad": (C.c_int, [C.c_int] * 6 + [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int64, _P]),
    "sntc_bias_grad_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "sntc_bias_grad": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int, _P, C.c_int64, _P]),
    "sntc_act_backward": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P]),
    "sntc_gate_forward": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "sntc_gate_backward": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P]),
    "sntc_axpy": (C.c_int, [_P, _P, C.c_float, C.c_int64, _P]),
    "sntc_noise_add": (C.c_int, [_P, C.c_int64, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "sntc_sumsq": (C.c_int, [_P, C.c_int64, _P, _P]),
    "sntc_two_layer_hidden": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "sntc_gdn_reparam_forward": (C.c_int, [_P, C.c_int64, C.c_float, C.c_float, _P, _P]),
    "sntc_gdn_reparam_backward": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P, _P]),
    "sntc_noisy_normal": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P]),
    "sntc_noisy_factorized": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, _P, _P]),
    "sntc_gdn_apply": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P]),
    "sntc_gdn_backward_prep": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P]),
    "sntc_gdn_backward_finish": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, _P, _P]),
    "sntc_small_matmul": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "sntc_transpose_last2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "sntc_prior_record_floats": (C.c_int, [_P]),
    "sntc_prior_update": (C.c_int, [_P, _P, _P, _P, _P]),
    "sntc_prior_param_grad": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load libsntc_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C shallow-ntc_amd/csrc`.  There is no CPU fallback for the hot path.")
    # torch first: its wheel carries its own libamdhip64, and the library below must bind to THAT runtime (the one that owns
    # torch's streams and allocations).  Loaded the other way round, /opt/rocm's copy comes in first, torch then brings a
    # second HIP runtime into the process, and one of them sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # A GPU that other processes or streams share cannot promise stream-K's workers co-residency (include/sntc.h, "Stream-K health"):
    # SNTC_STATIC_SCHEDULES=1 makes every launch of this process take the one-workgroup-per-tile / split-K schedules from the
    # start (same bits), instead of finding out through a timed-out hand-off and check_conv_status().
    if os.environ.get("SNTC_STATIC_SCHEDULES", "") not in ("", "0"):
        lib.sntc_conv_set_stream_k(0)
    return lib


def last_error() -> str:
    return load().sntc_last_error().decode("utf-8", "replace")


def check(rc: int):
    if rc != OK:
        msg = last_error()
        raise (NonFiniteError if rc == ERR_NONFINITE else SntcError)(rc, msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


def device_count() -> int:
    return load().sntc_device_count()


def require_gpu():
    """Fail loudly when the hot path is asked to run without a gfx950 device."""
    if device_count() < 1:
        raise SntcError(ERR_NO_DEVICE, "no HIP device visible: the shallow-ntc hot path only runs on MI355X (gfx950)")


def device_arch(device=0) -> str:
    buf = C.create_string_buffer(256)
    check(load().sntc_device_arch(device, buf, 256))
    return buf.value.decode()
