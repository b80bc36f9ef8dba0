"""ctypes binding of ``libtheia_hip.so`` (C ABI declared in ``include/theia_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or does not load, every entry point
raises.  ``torch`` is imported first so that the library binds to the HIP runtime torch already loaded
(same soname ``libamdhip64.so.7``) instead of pulling a second runtime into the process.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must precede CDLL: provides libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtheia_hip.so")

F32, BF16, FP8 = 0, 1, 2
ERR_UNSUPPORTED = -3  # THEIA_ERR_UNSUPPORTED
ABI_VERSION = 12
COMM_ID_BYTES = 128  # THEIA_COMM_ID_BYTES
ACT_NONE, ACT_GELU, ACT_RELU, ACT_MUL_DGELU, ACT_MUL_DRELU = 0, 1, 2, 3, 4
MAX_TAPS = 9


class RowMap(C.Structure):
    _fields_ = [
        ("ntaps", C.c_int32),
        ("dy", C.c_int32 * MAX_TAPS),
        ("dx", C.c_int32 * MAX_TAPS),
        ("wslot", C.c_int32 * MAX_TAPS),
        ("rows_h", C.c_int32), ("rows_w", C.c_int32),
        ("in_h", C.c_int32), ("in_w", C.c_int32),
        ("in_sy", C.c_int32), ("in_sx", C.c_int32),
        ("in_c", C.c_int32),
        ("out_w", C.c_int32), ("out_sy", C.c_int32), ("out_sx", C.c_int32), ("out_y0", C.c_int32), ("out_x0", C.c_int32),
        ("in_batch_stride", C.c_int64), ("in_offset", C.c_int64),
        ("out_batch_stride", C.c_int64), ("out_offset", C.c_int64),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p),
        ("bias", C.c_void_p), ("resid", C.c_void_p), ("aux_in", C.c_void_p), ("aux_out", C.c_void_p),
        ("rowtab", C.c_void_p), ("rowtab_period", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("ldw", C.c_int32), ("ldo", C.c_int32), ("act", C.c_int32),
        ("map", RowMap),
        ("tile", C.c_int32), ("reserved", C.c_int32),
        ("ln_sums", C.c_void_p),
        ("a_scale_inv", C.c_void_p), ("w_scale_inv", C.c_void_p),
        ("out8", C.c_void_p), ("out8_scale", C.c_void_p),
    ]


class CastJob(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p),
        ("d0", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32), ("dst_f32", C.c_int32),
        ("s0", C.c_int64), ("s1", C.c_int64), ("s2", C.c_int64),
        ("t0", C.c_int64), ("t1", C.c_int64),
        ("first_block", C.c_int64), ("tiles1", C.c_int32), ("tiles2", C.c_int32), ("tile1", C.c_int32), ("tile2", C.c_int32),
    ]


class Q8Out(C.Structure):  # theia_q8_out_t
    _fields_ = [("out", C.c_void_p), ("scale", C.c_void_p), ("amax", C.c_void_p)]


class QuantJob(C.Structure):  # theia_quant_job_t
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("scale", C.c_void_p), ("amax", C.c_void_p), ("n", C.c_int64),
                ("first_block", C.c_int32), ("pad_", C.c_int32)]


class WgradFinishJob(C.Structure):
    """theia_wgrad_finish_job_t"""
    _fields_ = [("slabs", C.c_void_p), ("out", C.c_void_p), ("bias_slabs", C.c_void_p), ("bias_out", C.c_void_p), ("sn", C.c_int64),
                ("N", C.c_int32), ("C", C.c_int32), ("accumulate", C.c_int32), ("bias_accumulate", C.c_int32)]


WGRAD_FINISH_GROUP_MAX = 4


class WgradArgs(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("a", C.c_void_p), ("slabs", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("ldo", C.c_int32), ("kslots", C.c_int32), ("splits", C.c_int32),
        ("map", RowMap),
        ("bias_slabs", C.c_void_p), ("bias_out", C.c_void_p), ("bias_accumulate", C.c_int32), ("defer_bias_reduce", C.c_int32),
    ]


_SIGNATURES = {
    # name: (restype, argtypes)
    "theia_abi_version": (C.c_int, []),
    "theia_last_error": (C.c_char_p, []),
    "theia_dtype_size": (C.c_int, [C.c_int]),
    "theia_gemm_nt": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_void_p]),
    "theia_gemm_nt_tile": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "theia_gemm_nt_plan": (C.c_int, [C.POINTER(GemmArgs), C.c_int]),
    "theia_gemm_wgrad": (C.c_int, [C.POINTER(WgradArgs), C.c_int, C.c_void_p]),
    "theia_gemm_wgrad_group": (C.c_int, [C.POINTER(WgradArgs), C.c_int, C.c_int, C.c_void_p]),
    "theia_wgrad_group_splits": (C.c_int, [C.c_int, C.c_int]),
    "theia_wgrad_tiles": (C.c_int, [C.c_int, C.c_int]),
    "theia_wgrad_fuses_bias": (C.c_int, [C.POINTER(WgradArgs), C.c_int]),
    "theia_wgrad_splits": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "theia_wgrad_splits_taps": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "theia_gemm_wgrad_plan": (C.c_int, [C.POINTER(WgradArgs), C.c_int]),
    "theia_set_compute_cus": (C.c_int, [C.c_int]),
    "theia_get_compute_cus": (C.c_int, []),
    "theia_set_gemm_schedule": (C.c_int, [C.c_int]),
    "theia_get_gemm_schedule": (C.c_int, []),
    "theia_wgrad_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_int, C.c_void_p]),
    "theia_wgrad_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "theia_wgrad_finish_group": (C.c_int, [C.POINTER(WgradFinishJob), C.c_int, C.c_int, C.c_void_p]),
    "theia_colsum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "theia_quantize_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "theia_fp8_update_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "theia_layernorm_fwd_q8": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_float, C.c_int, C.POINTER(Q8Out), C.c_void_p]),
    "theia_layernorm_bwd_q8": (C.c_int, [C.c_void_p] * 10 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(Q8Out), C.c_void_p]),
    "theia_layernorm_chw_fwd_sums_q8": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int64, C.c_float, C.c_int, C.POINTER(Q8Out), C.c_void_p]),
    "theia_layernorm_chw_bwd_colsum_q8": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                   C.POINTER(Q8Out), C.c_void_p]),
    "theia_distill_loss_bwd_q8": (C.c_int, [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int64, C.c_int, C.POINTER(Q8Out), C.c_void_p]),
    "theia_quantize_fp8_batch_plan": (C.c_int64, [C.c_void_p, C.c_int]),
    "theia_quantize_fp8_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "theia_colsum_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "theia_cast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "theia_cast_batch_plan": (C.c_int64, [C.c_void_p, C.c_int]),
    "theia_cast_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "theia_cast_transpose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "theia_cast_permute3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int, C.c_void_p]),
    "theia_unpermute3_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int, C.c_void_p]),
    "theia_transpose_acc_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_transpose_acc2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_patchify_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_patchify_u8_hw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_write_cls": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_write_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_layernorm_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "theia_layernorm_bwd": (C.c_int, [C.c_void_p] * 10 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "theia_layernorm_chw_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "theia_layernorm_chw_fwd_sums": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "theia_layernorm_chw_bwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_layernorm_chw_bwd_colsum": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_layernorm_chw_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "theia_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_attention_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_attention_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "theia_distill_loss_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "theia_distill_loss_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "theia_distill_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "theia_distill_loss_fwd_t": (C.c_int, [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "theia_distill_loss_bwd_t": (C.c_int, [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "theia_token_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_resize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_feature_ingest_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_feature_norm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "theia_add_inplace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "theia_fill_zero": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "theia_scatter_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "theia_adamw_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 8 + [C.c_void_p]),
    "theia_adamw_step_scaled": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 7 + [C.c_void_p, C.c_void_p]),
    "theia_adamw_step_dev": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "theia_upcast_scale_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "theia_comm_unique_id": (C.c_int, [C.c_void_p]),
    "theia_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "theia_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "theia_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "theia_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "theia_comm_destroy": (C.c_int, [C.c_void_p]),
    "theia_grad_sumsq_blocks": (C.c_int, []),
    "theia_grad_sumsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "theia_grad_clip_coef": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "theia_probe_tr16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib: Optional[C.CDLL] = None


class TheiaNativeError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the library; raises if it is not built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TheiaNativeError(
                f"{LIB_PATH} is missing: build it with `python -m theia_amd.build` (hipcc --offload-arch=gfx950). "
                "theia_amd has no CPU/eager fallback.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise TheiaNativeError(f"failed to load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.theia_abi_version() != ABI_VERSION:
            raise TheiaNativeError("libtheia_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().theia_last_error()
        raise TheiaNativeError(f"{what}: rc={rc}: {msg.decode() if msg else ''}")


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    if t == torch.float8_e4m3fn:  # THEIA_FP8: theia_gemm_nt operands only
        return FP8
    raise TypeError(f"unsupported dtype {t}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream on the current device.  Every C-ABI launch asks for it (~500 times per train step):
    ``torch.cuda.current_stream()`` builds a Stream object through three Python layers (~9 us per call, 2 ms of host time per step in the
    cProfile of tools/host_profile.py); the raw getter is a single C call."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
