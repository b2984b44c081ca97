"""ctypes binding of libsage_gfx950.so (include/sage_gfx950.h).

The library is the product: if it cannot be loaded this module raises -- there is no eager /
PyTorch / CPU fallback anywhere in ``sageattention_amd``.  Build it with
``python __graft_entry__.py`` (or ``make -C sageattention_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAGE_GFX950_LIB: load another build of the same library (A/B variants of tools/build_variants.sh)
LIB_PATH = os.environ.get("SAGE_GFX950_LIB") or os.path.join(_HERE, "libsage_gfx950.so")

# mirrors of the header's constants
ABI_VERSION = 21
DTYPE_F16, DTYPE_BF16 = 0, 1
GRAN_PER_BLOCK, GRAN_PER_WARP, GRAN_PER_THREAD = 1, 2, 3
GRAN_KBLK128 = 0x100          # OR-ed into the attention call's granularity: k scale groups of 128 keys
QSTYLE_TRITON, QSTYLE_CUDA, QSTYLE_TRITON_THREAD = 0, 1, 2
PV_ACCUM_SINGLE, PV_ACCUM_TWO_LEVEL, PV_ACCUM_TRITON = 0, 1, 2   # 2: FP16 PV, the reference's Triton kernel form
MASK_BOOL, MASK_F16, MASK_BF16 = 1, 2, 3
ATTR_FP8_EXACT_SCORES, ATTR_FORCE_PERSISTENT, ATTR_FP8_FOLDED_SCORES = 1, 2, 4


class SageLaunchAttr(ctypes.Structure):
    """``SageLaunchAttr`` of include/sage_gfx950.h: the launch attributes an attention entry point takes as its last argument."""
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("launch_ws", c_void_p), ("launch_ws_bytes", c_int64),
                ("grid_out", c_void_p), ("trace", c_void_p), ("trace_wgs", ctypes.c_int32), ("reserved", ctypes.c_int32)]


def launch_attr(launch_ws=None, folded_scores: bool = False, force_persistent: bool = False, grid_out=None, trace=None, trace_wgs: int = 0):
    """A ``SageLaunchAttr`` (or None when every field is at its default).  ``launch_ws``: a zeroed int32 CUDA tensor of
    ``sage_attn_launch_ws_bytes()`` bytes; the caller keeps it (and the returned struct) alive until the C call has returned.
    ``grid_out``: a ``ctypes.c_int32`` that receives the number of workgroups launched."""
    if launch_ws is None and not folded_scores and grid_out is None and trace is None:
        return None
    a = SageLaunchAttr()
    a.struct_bytes = ctypes.sizeof(SageLaunchAttr)
    a.flags = (ATTR_FP8_FOLDED_SCORES if folded_scores else 0) | (ATTR_FORCE_PERSISTENT if force_persistent else 0)
    if launch_ws is not None:
        a.launch_ws = launch_ws.data_ptr()
        a.launch_ws_bytes = launch_ws.numel() * launch_ws.element_size()
    if grid_out is not None:
        a.grid_out = ctypes.addressof(grid_out)
    if trace is not None:
        a.trace = trace.data_ptr()
        a.trace_wgs = int(trace_wgs)
    a._keep = (launch_ws, grid_out, trace)      # (the struct holds raw addresses)
    return a


def attr_arg(a):
    """The ``attr`` argument of an attention entry point: NULL or a pointer to the struct."""
    return None if a is None else ctypes.byref(a)

# every symbol include/sage_gfx950.h declares: name -> (restype, argtypes)
_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
SYMBOLS = {   # (the trailing _P of every sage_attn_* entry point is `const SageLaunchAttr *attr`, nullable)
    "sage_abi_version": (c_int, []),
    "sage_last_error": (ctypes.c_char_p, []),
    "sage_debug_work_order_plan": (c_int, [_I, _I, _L, _I, _I, _I, _P, _P, _P]),
    "sage_debug_work_item": (c_int, [_I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sage_work_order": (c_int, []),
    "sage_set_work_order": (None, [_I]),
    "sage_v_image_bytes": (c_int64, [_I, _I, _L]),
    "sage_quant_qk_int8": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L,
                                   _I, _I, _I, _I, _I, _F, _I, _P]),
    "sage_quant_qk_int8_varlen": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L,
                                          _I, _F, _I, _P]),
    "sage_stats_ws_floats": (c_int64, [_I, _I, _I, _I]),
    "sage_channel_mean": (c_int, [_P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _I, _P]),
    "sage_channel_mean_varlen": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _L, _I, _P]),
    "sage_prep_v_fp8": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _F, _I, _P]),
    "sage_prepass_ws_floats": (c_int64, [_I, _I, _I, _I]),
    "sage_prepass_sync_words": (c_int64, [_I, _I]),
    "sage_prepass_max_seqlen": (c_int, []),
    "sage_prepass_max_seqlen_stream": (c_int, [_P]),
    "sage_host_word_alloc": (c_int, [_P, _P]),
    "sage_host_word_free": (c_int, [_P]),
    "sage_prepass_failed_heads": (c_int, [_P, _I, _I, _P]),
    "sage_debug_prepass_fail": (None, [_I]),
    "sage_prepass_kv": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _F, _I, _I, _P, _P]),
    "sage_prepass_kv_varlen": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                       _L, _L, _L, _L, _L, _L, _I, _P, _P]),
    "sage_debug_spin": (c_int, [_I, _I, _P]),
    "sage_prep_v_f16": (c_int, [_P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _I, _P]),
    "sage_prep_v_f16_varlen": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _I, _P]),
    "sage_attn_launch_ws_bytes": (c_int64, []),
    "sage_attn_qk_int8_pv_f8": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                        _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _F, _I, _I, _P, _P]),
    "sage_attn_qk_int8_pv_f16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                         _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _F, _I, _I, _P, _P]),
    "sage_attn_qk_int8_pv_f16_vrows": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                               _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _F, _I, _I, _P, _P]),
    "sage_attn_qk_int8_pv_f16_masked": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I,
                                                _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _I, _P, _P]),
    "sage_attn_qk_int8_pv_f16_varlen": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                                _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_attn_fused_q_pv_f8": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                        _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_attn_fused_q_pv_f16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                         _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_attn_fused_q_pv_f16_vrows": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                               _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _P, _P]),
    "sage_attn_fused_qblock_pv_f16_vrows": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                                    _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _P, _P]),
    "sage_varlen_plan_max_seqs": (c_int, []),
    "sage_varlen_plan": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P]),
    "sage_debug_varlen_items": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "sage_attn_fused_qblock_pv_f16": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                              _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_attn_fused_qblock_pv_f16_varlen": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                                     _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_attn_fused_q_pv_f8_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I,
                                              _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_attn_fused_q_pv_f16_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I,
                                               _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _F, _I, _I, _P, _P]),
    "sage_merge_states": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _I, _I, _P]),
    "sage_merge_split": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _L, _L, _I, _P]),
}


class SageLibraryError(RuntimeError):
    """libsage_gfx950.so is missing or does not match include/sage_gfx950.h."""


class SageKernelError(RuntimeError):
    """A C-ABI call returned a negative status (SAGE_EINVAL / SAGE_ELAUNCH)."""


_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library (once); raise loudly if it is absent -- no fallback path exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SageLibraryError(
            f"{LIB_PATH} not found: the gfx950 HIP extension has not been built. "
            "Run `python __graft_entry__.py` (or `make -C sageattention_amd/csrc`). "
            "sageattention_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SageLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.sage_abi_version()
    if got != ABI_VERSION:
        raise SageLibraryError(f"ABI version mismatch: library {got}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sage_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")          # the reference raises ValueError for bad arguments
        raise SageKernelError(f"{what} failed ({rc}): {msg}")
