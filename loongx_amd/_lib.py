"""ctypes binding of liblx_amd.so (include/lx.h).

This is the FFI stub a maintainer of the reference would add.  The product path has NO
fallback: if the HIP library is missing or fails to load, importing raises; kernels are never replaced by
torch ops.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7; liblx_amd.so needs the SAME runtime instance (streams and device
# pointers are shared with torch), so torch -- and with it its HIP runtime -- must be loaded before our library.
import torch  # noqa: F401  (import order is load-bearing)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LX_AMD_LIB", os.path.join(_HERE, "lib", "liblx_amd.so"))

LX_EPI_STORE_BF16, LX_EPI_STORE_F32, LX_EPI_RESID_F32, LX_EPI_GELU, LX_W_TILED, LX_EPI_SPLIT_BF16 = 0, 1, 2, 0x100, 0x200, 0x400
LX_EPI_STORE_FP8, LX_OPERANDS_FP8, LX_EPI_QKV, LX_OPERANDS_F16 = 3, 0x800, 0x1000, 0x2000
LX_GEMM_MAX_GROUP = 4


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p),
                ("gate", C.c_void_p), ("lora_t", C.c_void_p), ("lora_up", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32),
                ("rows_per_batch", C.c_int32), ("gate_ld", C.c_int32),
                ("lora_r", C.c_int32), ("lora_ldt", C.c_int32), ("lora_mod_cols", C.c_int32),
                ("lora_toff_max", C.c_int32), ("epilogue", C.c_int32), ("gelu_col_start", C.c_int32),
                ("lora_nsplit", C.c_int32), ("lora_split_stride", C.c_int32),
                ("k_segs", C.c_int32), ("a_lo_off", C.c_int32), ("c_lo_off", C.c_int32), ("out_scale", C.c_float),
                ("col_scale", C.c_void_p),          # (LX_OPERANDS_F16: the same slot is f16_ovf -- a union in lx.h)
                ("qkv_norm_q", C.c_void_p), ("qkv_norm_k", C.c_void_p), ("qkv_rope", C.c_void_p), ("qkv_vt", C.c_void_p),
                ("qkv_k", C.c_void_p),
                ("qkv_d", C.c_int32), ("qkv_vt_ld", C.c_int32), ("qkv_vt_pos0", C.c_int32), ("qkv_k_ld", C.c_int32),
                ("qkv_q8", C.c_void_p), ("qkv_k8", C.c_void_p), ("qkv_vt8", C.c_void_p),
                ("qkv_ld8", C.c_int32), ("qkv_q_scale", C.c_float), ("qkv_k_scale", C.c_float), ("qkv_v_scale", C.c_float)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("VT", C.c_void_p), ("O", C.c_void_p),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldo", C.c_int32), ("vt_ld", C.c_int32),
                ("q_col", C.c_int32), ("k_col", C.c_int32), ("o_col", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("n_seg", C.c_int32),
                ("seg_row0", C.c_int32 * 3), ("seg_len", C.c_int32 * 3), ("seg_vt0", C.c_int32 * 3),
                ("bias", (C.c_float * 3) * 3), ("scale", C.c_float), ("n_qseg", C.c_int32), ("flags", C.c_int32),
                ("qseg_mask", C.c_int32), ("f16_ovf", C.c_void_p)]


LX_ATTN_Q_LOG2, LX_ATTN_BOUNDED, LX_ATTN_INVARIANT, LX_ATTN_O_F16, LX_ATTN_PREFER_4WAVE, LX_ATTN_P_EXP2 = 1, 2, 4, 8, 16, 32


class AttnF32Desc(C.Structure):
    _fields_ = [("QKV", C.c_void_p), ("ld", C.c_int32), ("q_col", C.c_int32), ("k_col", C.c_int32), ("v_col", C.c_int32),
                ("O", C.c_void_p), ("ldo", C.c_int32), ("o_col", C.c_int32), ("o_lo_off", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("n_seg", C.c_int32),
                ("seg_row0", C.c_int32 * 3), ("seg_len", C.c_int32 * 3),
                ("bias", (C.c_float * 3) * 3), ("scale", C.c_float)]


class LnSeg(C.Structure):
    _fields_ = [("row0", C.c_int32), ("n_rows", C.c_int32), ("rows_per_batch", C.c_int32), ("_pad", C.c_int32),
                ("shift", C.c_void_p), ("scale", C.c_void_p)]


class QkvSeg(C.Structure):
    _fields_ = [("row0", C.c_int32), ("rows_per_batch", C.c_int32), ("vt_pos0", C.c_int32), ("_pad", C.c_int32),
                ("wq", C.c_void_p), ("wk", C.c_void_p), ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p)]


_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGS = {
    "lx_version": (C.c_int, []),
    "lx_last_error": (C.c_char_p, []),
    "lx_device_arch": (C.c_int, [C.c_char_p, _Z]),
    "lx_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), _I, _P]),
    "lx_gemm_reload_env": (None, []),
    "lx_gemm_workspace_bytes": (_Z, []),
    "lx_gemm_bf16_ws": (C.c_int, [C.POINTER(GemmDesc), _I, _P, _Z, _P]),
    "lx_gemm_workspace_status": (C.c_int, [_P, _P]),
    "lx_lora_down": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lx_lora_down_f16": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lx_lora_down_terms": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), _I, _P, _I, _I, _I, _I, _I, _P]),
    "lx_linear_skinny": (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "lx_timestep_embed": (C.c_int, [_P, _P, _I, _I, _P]),
    "lx_rope_table": (C.c_int, [_P, _I, _I, _I, _I, C.c_double, _P, _P, _P]),
    "lx_ln_modulate": (C.c_int, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _F, _P]),
    "lx_ln_modulate_segs": (C.c_int, [_P, _I, C.POINTER(LnSeg), _I, _I, _P, _I, _I, _F, _P]),
    "lx_ln_modulate_f16_segs": (C.c_int, [_P, _I, C.POINTER(LnSeg), _I, _I, _P, _I, _I, _F, _P, _P]),
    "lx_ln_modulate_lora_segs": (C.c_int, [_P, _I, C.POINTER(LnSeg), _I, _I, _P, _I, _I, _F, _P, _I, _P, _I, _I, _I, _P]),
    "lx_ln_modulate_lora_f16_segs": (C.c_int, [_P, _I, C.POINTER(LnSeg), _I, _I, _P, _I, _I, _F, _P, _I, _P, _I, _I, _I, _P, _P]),
    "lx_qkv_prep_segs": (C.c_int, [_P, _I, _I, _I, _I, C.POINTER(QkvSeg), _I, _I, _I, _F, _P, _I, _P]),
    "lx_qkv_prep_f16in_segs": (C.c_int, [_P, _I, _I, _I, _I, C.POINTER(QkvSeg), _I, _I, _I, _F, _P, _I, _P]),
    "lx_qkv_prep": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _P]),
    "lx_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), _P]),
    "lx_attn_last_kernel": (C.c_int, []),
    "lx_qkv_prep_fp8_segs": (C.c_int, [_P, _I, _I, _I, _I, C.POINTER(QkvSeg), _I, _I, _I, _F, _P, _P, _I, _P, _I, _F, _F, _F, _P]),
    "lx_qkv_prep_fp8_f16in_segs": (C.c_int, [_P, _I, _I, _I, _I, C.POINTER(QkvSeg), _I, _I, _I, _F, _P, _P, _I, _P, _I, _F, _F, _F, _P]),
    "lx_attn_fwd_fp8": (C.c_int, [C.POINTER(AttnDesc), _F, _F, _P]),
    "lx_split_bf16": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "lx_ln_modulate_split_segs": (C.c_int, [_P, _I, C.POINTER(LnSeg), _I, _I, _P, _I, _I, _I, _F, _P]),
    "lx_qkv_prep_f32_segs": (C.c_int, [_P, _I, _I, _I, C.POINTER(QkvSeg), _I, _I, _I, _F, _P]),
    "lx_attn_fwd_f32": (C.c_int, [C.POINTER(AttnF32Desc), _P]),
    "lx_qkv_prep_split_segs": (C.c_int, [_P, _I, _I, _I, _I, C.POINTER(QkvSeg), _I, _I, _I, _F, _P, _I, _I, _I, _I, _P, _I, C.c_longlong, _P]),
    "lx_attn_fwd_split": (C.c_int, [C.POINTER(AttnDesc), _I, C.c_longlong, _I, _P]),
    "lx_groupnorm_workspace_bytes": (_Z, [_I, _I, _I]),
    "lx_groupnorm_silu": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _F, _I, _P, _P, _Z, _P]),
    "lx_im2col3x3": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "lx_softmax_rows": (C.c_int, [_P, _I, _F, _P, _I, _I, _I, _P]),
    "lx_ln_modulate_fp8_segs": (C.c_int, [_P, _I, C.POINTER(LnSeg), _I, _I, _P, _I, _P, _I, _F, _I, _F, _P]),
    "lx_convert_fp8": (C.c_int, [_P, _I, _I, _P, _I, _F, _I, _I, _P]),
    "lx_lora_down_fp8": (C.c_int, [_P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lx_euler_step": (C.c_int, [_P, _P, _I, _F, _Z, _P]),
    "lx_convert": (C.c_int, [_P, _I, _P, _I, _Z, _P]),
    "lx_s4_scan": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "lx_s4_conv": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "lx_chanmix": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lx_pyramid_pool": (C.c_int, [_P, _P, _I, _I, _I, C.POINTER(C.c_int), _I, _I, _I, _P]),
    "lx_layernorm_relu": (C.c_int, [_P, _P, _P, _I, _I, _F, _P]),
    "lx_linear_f32": (C.c_int, [_P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lx_chan_gemm_f32": (C.c_int, [_P, C.c_long, _I, _P, _I, _P, _P, C.c_long, _I, _I, _I, _I, _I, _I, _P, _P]),
    "lx_duan_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "lx_duan_fwd": (C.c_int, [_P] * 11 + [_I, _I, _I, _I, _F, _I, _P, _Z, _P]),
}

EXPORTS = tuple(_SIGS)


class LxError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"liblx_amd.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                          f"g.build()'` or loongx_amd/csrc/build.sh -- there is no CPU/torch fallback for the hot path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError if the .so is stale: fail loudly
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(status: int, what: str = "") -> None:
    if status != 0:
        raise LxError(f"{what or 'lx call'} failed ({status}): {lib.lx_last_error().decode()}")
