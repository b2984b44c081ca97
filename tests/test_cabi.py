"""CPU: the C-ABI library loads, exports exactly what include/sage_gfx950.h declares, and rejects bad
arguments before touching a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

import util  # noqa: F401  (sys.path)
from sageattention_amd import _cabi

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sage_gfx950.h")


def declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"SAGE_API\s+[\w\s\*]+?\b(sage_\w+)\s*\(", txt)))


def test_binding_covers_header():
    assert declared_symbols() == sorted(_cabi.SYMBOLS)


def test_library_exports_every_symbol():
    assert os.path.exists(_cabi.LIB_PATH), "run `python __graft_entry__.py` first"
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} not exported"
    assert _cabi.load().sage_abi_version() == _cabi.ABI_VERSION


def test_argument_errors_do_not_need_a_gpu():
    lib = _cabi.load()
    buf = ctypes.create_string_buffer(4096)
    p = (ctypes.addressof(buf) + 15) & ~15
    # head_dim 96 is rejected (the Python layer pads, core.py:260-271)
    rc = lib.sage_quant_qk_int8(p, None, p, p, 1, 1, 16, 96, 0, 0, 96, 0, 0, 96, 0, 0, 128, 128, 1, 0, 0, 1.0, 0, None)
    assert rc == -1 and b"head_dim" in lib.sage_last_error()
    # misaligned pointer
    rc = lib.sage_quant_qk_int8(p + 2, None, p, p, 1, 1, 16, 64, 0, 0, 64, 0, 0, 64, 0, 0, 128, 128, 1, 0, 0, 1.0, 0, None)
    assert rc == -1 and b"aligned" in lib.sage_last_error()
    # Hq not divisible by Hkv
    rc = lib.sage_attn_qk_int8_pv_f16(p, p, p, p, None, p, p, None, 1, 3, 2, 16, 16, 64, 0, 0, 64, 0, 0, 64, 0, 0, 64, 0, 1, 128, 1.0, 1, 0, None, None)
    assert rc == -1 and b"divisible" in lib.sage_last_error()
    with pytest.raises(ValueError):
        _cabi.check(rc, "x")
    # fused-Q attention: q dtype code, kv head count, strides
    rc = lib.sage_attn_fused_q_pv_f8(p, p, p, p, None, p, p, None, 1, 2, 2, 16, 16, 128, 0, 0, 128, 0, 0, 128, 0, 0, 128, 0, 1.0, 7, 0, None, None)
    assert rc == -1 and b"q_dtype" in lib.sage_last_error()
    rc = lib.sage_attn_fused_q_pv_f8(p, p, p, p, None, p, p, None, 1, 2, 2, 16, 16, 128, 0, 0, 132, 0, 0, 128, 0, 0, 128, 0, 1.0, 0, 0, None, None)
    assert rc == -1 and b"q strides" in lib.sage_last_error()
    # LSE merge: head_dim must be a multiple of 8, pointers non-null
    rc = lib.sage_merge_states(p, p, p, p, None, 1, 1, 4, 12, 0, 0, 12, 0, 0, 0, 0, 0, None)
    assert rc == -1 and b"multiple of 8" in lib.sage_last_error()
    rc = lib.sage_merge_states(p, None, p, p, None, 1, 1, 4, 16, 0, 0, 16, 0, 0, 0, 0, 0, None)
    assert rc == -1 and b"null" in lib.sage_last_error()
    # varlen attention needs its prefix arrays
    rc = lib.sage_attn_qk_int8_pv_f16_varlen(p, p, p, p, p, p, None, None, None, None, None, None, None, 0, 1, 16, 2, 2, 64, 128, 64, 128, 64, 128, 64,
                                             0, 1.0, 1, 0, None, None)
    assert rc == -1 and b"varlen" in lib.sage_last_error()
    # a work list comes with its header and a positive bound
    rc = lib.sage_attn_qk_int8_pv_f16_varlen(p, p, p, p, p, p, p, p, p, p, None, p, None, 4, 1, 16, 2, 2, 64, 128, 64, 128, 64, 128, 64,
                                             0, 1.0, 1, 0, None, None)
    assert rc == -1 and b"work list" in lib.sage_last_error()
    # the varlen plan: sequence count, and the work list needs the attention kernel's block sizes
    rc = lib.sage_varlen_plan(p, p, 5000, 0, 128, 64, 0, 8, 8, 128, 0, None, p, p, None, 0, None, None, 0, None, None)
    assert rc == -1 and b"nseq" in lib.sage_last_error()
    rc = lib.sage_varlen_plan(p, p, 4, 0, 64, 64, 0, 8, 8, 128, 0, None, p, p, p, 16, None, None, 0, p, None)
    assert rc == -1 and b"work list" in lib.sage_last_error()
    rc = lib.sage_varlen_plan(p, p, 4, 100, 128, 64, 0, 8, 8, 128, 0, None, p, p, p, 0, p, p, 8, p, None)       # a work list without its capacity
    assert rc == -1 and b"capacities" in lib.sage_last_error()
    # the work-order debug view checks every argument (a zero block count used to divide by zero)
    h, r = ctypes.c_int(), ctypes.c_int()
    assert lib.sage_debug_work_item(0, 8, 8, 0, 2, 0, 0, ctypes.byref(h), ctypes.byref(r)) == -1
    assert lib.sage_debug_work_item(0, 8, 8, 4, 2, 0, 9, ctypes.byref(h), ctypes.byref(r)) == -1
    # the one-launch varlen pre-pass refuses more slabs per head than can wait for each other
    rc = lib.sage_prepass_kv_varlen(p, p, p, p, p, p, p, p, p, p, p, p, p, 4, 512 * 200, 512, 204, 2, 128, 256, 128, 256, 128, 128, 512 * 200 * 128,
                                    0, None, None)
    assert rc == -1


def test_launch_attributes_are_checked_arguments():
    """SageLaunchAttr travels with the call it belongs to: workspace size / alignment, unknown flags and a force flag without a workspace
    are refused BEFORE the tensor arguments are looked at (so no GPU is needed to see it); a shorter struct (an older caller) is accepted;
    NULL is the default.  The library exports no setter any more."""
    lib = _cabi.load()
    n = int(lib.sage_attn_launch_ws_bytes())
    assert n == 32 * 128
    assert not hasattr(lib, "sage_attn_launch_ws") and not hasattr(lib, "sage_debug_last_attn_grid")
    buf = ctypes.create_string_buffer(n + 256)
    p = (ctypes.addressof(buf) + 127) & ~127

    def call(attr):      # Hq not divisible by Hkv: the call fails its argument checks either way, never launches
        rc = lib.sage_attn_qk_int8_pv_f16(p, p, p, p, None, p, p, None, 1, 3, 2, 16, 16, 64, 0, 0, 64, 0, 0, 64, 0, 0, 64, 0, 1, 128, 1.0, 1, 0, None,
                                          None if attr is None else ctypes.byref(attr))
        return rc, lib.sage_last_error()

    def attr(**kw):
        a = _cabi.SageLaunchAttr()
        a.struct_bytes = ctypes.sizeof(a)
        for k_, v_ in kw.items():
            setattr(a, k_, v_)
        return a

    assert call(None) == (-1, lib.sage_last_error()) and b"divisible" in lib.sage_last_error()
    rc, msg = call(attr(launch_ws=p, launch_ws_bytes=n - 1))
    assert rc == -1 and b"launch workspace" in msg
    rc, msg = call(attr(launch_ws=p + 64, launch_ws_bytes=n))
    assert rc == -1 and b"launch workspace" in msg
    rc, msg = call(attr(flags=0x80))
    assert rc == -1 and b"flags" in msg
    rc, msg = call(attr(flags=_cabi.ATTR_FORCE_PERSISTENT))
    assert rc == -1 and b"FORCE_PERSISTENT" in msg
    rc, msg = call(attr(struct_bytes=4))
    assert rc == -1 and b"struct_bytes" in msg
    rc, msg = call(attr(struct_bytes=0, launch_ws=p, launch_ws_bytes=n))     # a caller that never set it: refused, not read as a full struct
    assert rc == -1 and b"struct_bytes" in msg
    rc, msg = call(attr(flags=_cabi.ATTR_FP8_EXACT_SCORES | _cabi.ATTR_FP8_FOLDED_SCORES))
    assert rc == -1 and b"both FP8 score forms" in msg
    for ok in (attr(launch_ws=p, launch_ws_bytes=n), attr(flags=_cabi.ATTR_FP8_EXACT_SCORES), attr(flags=_cabi.ATTR_FP8_FOLDED_SCORES), attr(struct_bytes=8),
               attr(struct_bytes=4096, launch_ws=p, launch_ws_bytes=n)):      # (a newer caller's longer struct: the bytes this library knows are read)
        rc, msg = call(ok)
        assert rc == -1 and b"divisible" in msg          # past the attribute checks


def test_v_image_bytes():
    lib = _cabi.load()
    assert lib.sage_v_image_bytes(128, 1, 10) == 10 * 128 * 64
    assert lib.sage_v_image_bytes(64, 0, 3) == 3 * 64 * 128


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_cabi, "_lib", None)
    monkeypatch.setattr(_cabi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_cabi.SageLibraryError, match="no CPU/PyTorch fallback"):
        _cabi.load()
