"""torch.library registration of the fused attention ops (the reference's ``sm80_compile.py`` /
``sm89_compile.py`` / ``sm90_compile.py`` convention): ``custom_op(..., mutates_args=("output",))`` plus
a fake (shape-only) implementation, so ``torch.compile`` can trace through ``sageattn`` in
non-fullgraph / no-cudagraph mode exactly as the reference documents (README.md:30, core.py:252-257).

The ops are thin: they forward data pointers, shapes and strides to the C ABI (``include/sage_gfx950.h``).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _cabi, _stream_cache


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dims(t: torch.Tensor, tensor_layout: int):
    # tensor_layout encoding of the reference's op boundary: 0 = NHD, 1 = HND (core.py:556)
    if tensor_layout == 1:
        B, H, L, D = t.shape
        return B, H, L, D, t.stride(0), t.stride(1), t.stride(2)
    B, L, H, D = t.shape
    return B, H, L, D, t.stride(0), t.stride(2), t.stride(1)


def _lse_alloc(query: torch.Tensor, tensor_layout: int, return_lse: int) -> torch.Tensor:
    B, H, L, _, _, _, _ = _dims(query, tensor_layout)
    if return_lse:
        return torch.empty((B, H, L), dtype=torch.float32, device=query.device)
    return torch.empty((0,), dtype=torch.float32, device=query.device)      # reference: torch::empty({0})


_PERSISTENT = os.environ.get("SAGE_PERSISTENT_LAUNCH", "1") != "0"      # 0: every attention launch leaves its order to the hardware
# twelve rounds of the 512 workgroups an MI355X holds (the library decides; this only spares smaller calls the memset)
_PERSISTENT_MIN_ITEMS = 6144
# FP8 PV score form of every call that does not say otherwise: "exact" -- the reference's formula, the default -- or the opt-in variant "folded"
# (include/sage_gfx950.h, SAGE_ATTR_FP8_FOLDED_SCORES)
_FP8_FOLDED = os.environ.get("SAGE_FP8_SCORES", "exact").lower() == "folded"


def fp8_folded(kwarg=None) -> bool:
    """``fp8_scores=`` of the FP8 entry points: "exact" (the default: exp2(fma(s, c, -m)) as the reference computes it, on every input), "folded"
    (an opt-in variant, 3-7 % faster, that rounds ``m + bias c`` once per (row, tile, k scale) -- re-rolls e4m3 roundings of P and degrades with the
    magnitude of q and k, include/sage_gfx950.h) or None = the process default (SAGE_FP8_SCORES, "exact" unless set)."""
    if kwarg is None:
        return _FP8_FOLDED
    if kwarg not in ("exact", "folded"):
        raise ValueError(f"fp8_scores must be 'exact' or 'folded' (got {kwarg!r})")
    return kwarg == "folded"


def attn_launch_ws(device: torch.device, is_causal, n_items: int, packed: bool = False) -> Optional[torch.Tensor]:
    """The launch workspace of an attention launch (``SageLaunchAttr.launch_ws``, include/sage_gfx950.h): a zero-on-entry counter block that
    lets a large launch run as a persistent launch over ticket queues -- NON-CAUSAL launches (2.2-2.8 % faster on the CogVideoX shape and on
    packed batches) and, ``packed``, the causal launch of ``sageattn_varlen`` over its work list (+2.9 % at C4); dense causal launches
    keep the hardware's dispatch.  Results do not depend on it.  Returns the tensor or None.  ``n_items``: 128-row query blocks x heads
    x batch of the call.  The launch returns the block to zero before it ends, so it is ONE block per (device, stream), zeroed when it is
    created (``_stream_cache``): no memset per call (rounds 4-5 allocated and zeroed a fresh one for every large call).
    A performance attribute only: SAGE_PERSISTENT_LAUNCH=0 switches it off."""
    if (is_causal and not packed) or not _PERSISTENT or n_items < _PERSISTENT_MIN_ITEMS:
        return None
    return _stream_cache.zeroed("attn_tickets", int(_cabi.load().sage_attn_launch_ws_bytes()) // 4, device)


class _Hooks:
    """Test / tool hooks of the attention launches.  They are read on every launch, so they live in ONE object that is empty unless a
    ``launch_hooks`` block is active -- not in module globals that a forgotten assignment leaves behind for every thread of the process."""
    __slots__ = ("grid_probe", "force_persistent", "trace_buf")

    def __init__(self):
        self.grid_probe = None           # a ctypes.c_int32: receives the number of workgroups of every attention launch (SageLaunchAttr.grid_out)
        self.force_persistent = False    # SAGE_ATTR_FORCE_PERSISTENT wherever a workspace travels along (tests: the ticket route from two rounds up)
        self.trace_buf = None            # tools/attn_trace.py: an int32 CUDA tensor of 16 words per logical workgroup (-DSAGE_ATTN_TRACE=1 builds only)


_hooks = _Hooks()


class launch_hooks:
    """``with ops.launch_hooks(grid_probe=c_int32(), force_persistent=True): ...`` -- tests and tools only.  The hooks apply to the attention
    launches issued inside the block (by any thread: they are process-wide while active) and are gone when it exits, also on an exception."""

    def __init__(self, grid_probe=None, force_persistent: bool = False, trace_buf=None):
        self._new = (grid_probe, bool(force_persistent), trace_buf)

    def __enter__(self):
        self._old = (_hooks.grid_probe, _hooks.force_persistent, _hooks.trace_buf)
        _hooks.grid_probe, _hooks.force_persistent, _hooks.trace_buf = self._new
        return _hooks

    def __exit__(self, *exc):
        _hooks.grid_probe, _hooks.force_persistent, _hooks.trace_buf = self._old
        return False


def attn_attr(device: torch.device, is_causal, n_items: int, folded_scores: bool = False, packed: bool = False):
    """The ``attr`` argument of an attention entry point for a call of ``n_items`` work items: NULL, or a ``SageLaunchAttr`` with the launch
    workspace (``attn_launch_ws``) and / or the folded FP8 score form.  The returned object references the workspace tensor: keep it until
    the C call has returned."""
    h = _hooks
    ws = attn_launch_ws(device, is_causal, n_items if not h.force_persistent else max(n_items, _PERSISTENT_MIN_ITEMS), packed)
    return _cabi.launch_attr(ws, folded_scores=folded_scores, force_persistent=h.force_persistent and ws is not None, grid_out=h.grid_probe,
                             trace=h.trace_buf, trace_wgs=0 if h.trace_buf is None else h.trace_buf.numel() // 16)


def attn_check(rc: int, what: str, attr, device: torch.device) -> None:
    """``_cabi.check`` for an attention entry point; a failed call's ticket block is forgotten (it may not be zero: the next call gets a fresh one)."""
    if rc != 0 and attr is not None and attr.launch_ws:
        _stream_cache.drop("attn_tickets", device)
    _cabi.check(rc, what)


def qk_int8_sv_f8_attn_impl(query: torch.Tensor, key: torch.Tensor, v_image: torch.Tensor, output: torch.Tensor,
                       query_scale: torch.Tensor, key_scale: torch.Tensor, value_scale: torch.Tensor,
                       value_mean: Optional[torch.Tensor], tensor_layout: int, is_causal: int, qk_quant_gran: int,
                       q_warp: int, sm_scale_log2: float, pv_accum: int, return_lse: int, folded_scores: bool = False) -> torch.Tensor:
    """INT8 QK^T + FP8 PV (replaces the sm89/sm90 ops, sm89_compile.py:5-146, sm90_compile.py:5-94).  ``folded_scores``: the opt-in folded
    FP8 score form instead of the reference's exact one (a gfx950 attribute, include/sage_gfx950.h)."""
    B, Hq, Lq, D, q_sb, q_sh, q_sl = _dims(query, tensor_layout)
    _, Hkv, Lk, _, k_sb, k_sh, k_sl = _dims(key, tensor_layout)
    _, _, _, _, o_sb, o_sh, o_sl = _dims(output, tensor_layout)
    lse = _lse_alloc(query, tensor_layout, return_lse)
    code = _cabi.DTYPE_F16 if output.dtype == torch.float16 else _cabi.DTYPE_BF16
    attr = attn_attr(query.device, is_causal, B * Hq * ((Lq + 127) // 128), folded_scores)      # (core resolves SAGE_FP8_SCORES; an explicit form wins)
    rc = _cabi.load().sage_attn_qk_int8_pv_f8(
        _p(query), _p(key), _p(v_image), _p(output), _p(lse) if return_lse else None, _p(query_scale), _p(key_scale),
        _p(value_scale), _p(value_mean), B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
        is_causal, qk_quant_gran, q_warp, float(sm_scale_log2), pv_accum, code,
        torch._C._cuda_getCurrentRawStream(output.device.index), _cabi.attr_arg(attr))
    attn_check(rc, "sage_attn_qk_int8_pv_f8", attr, query.device)
    return lse


# the plain function stays importable (eager callers use it directly, skipping the dispatcher); the registered op wraps it
qk_int8_sv_f8_attn = torch.library.custom_op("sageattention_gfx950::qk_int8_sv_f8_attn", qk_int8_sv_f8_attn_impl,
                                             mutates_args=("output",), device_types="cuda")


@qk_int8_sv_f8_attn.register_fake
def _(query, key, v_image, output, query_scale, key_scale, value_scale, value_mean, tensor_layout, is_causal,
      qk_quant_gran, q_warp, sm_scale_log2, pv_accum, return_lse, folded_scores=False):
    return _lse_alloc(query, tensor_layout, return_lse)


def qk_int8_sv_f16_attn_impl(query: torch.Tensor, key: torch.Tensor, v_image: torch.Tensor, output: torch.Tensor,
                        query_scale: torch.Tensor, key_scale: torch.Tensor, value_mean: Optional[torch.Tensor],
                        tensor_layout: int, is_causal: int, qk_quant_gran: int, q_warp: int, sm_scale_log2: float,
                        pv_accum: int, return_lse: int) -> torch.Tensor:
    """INT8 QK^T + FP16 PV (replaces the sm80 ops, sm80_compile.py:5-149, and the Triton forward).  ``v_image``: the gfx950 tile image
    (5-D, ``quant.prep_v_fp16`` / the fused pre-pass) -- or, as the reference's ops take it, the fp16 value tensor itself in the layout of
    ``key`` (4-D; ``sage_attn_qk_int8_pv_f16_vrows``: rows read in place, same bits)."""
    B, Hq, Lq, D, q_sb, q_sh, q_sl = _dims(query, tensor_layout)
    _, Hkv, Lk, _, k_sb, k_sh, k_sl = _dims(key, tensor_layout)
    _, _, _, _, o_sb, o_sh, o_sl = _dims(output, tensor_layout)
    lse = _lse_alloc(query, tensor_layout, return_lse)
    code = _cabi.DTYPE_F16 if output.dtype == torch.float16 else _cabi.DTYPE_BF16
    attr = attn_attr(query.device, is_causal, B * Hq * ((Lq + 127) // 128))
    if v_image.dim() == 4:
        assert v_image.dtype == torch.float16 and v_image.stride(-1) == 1, "value read in place: fp16, last dimension contiguous"
        _, _, _, _, v_sb, v_sh, v_sl = _dims(v_image, tensor_layout)
        rc = _cabi.load().sage_attn_qk_int8_pv_f16_vrows(
            _p(query), _p(key), _p(v_image), _p(output), _p(lse) if return_lse else None, _p(query_scale), _p(key_scale),
            _p(value_mean), B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl,
            is_causal, qk_quant_gran, q_warp, float(sm_scale_log2), pv_accum, code,
            torch._C._cuda_getCurrentRawStream(output.device.index), _cabi.attr_arg(attr))
        attn_check(rc, "sage_attn_qk_int8_pv_f16_vrows", attr, query.device)
        return lse
    rc = _cabi.load().sage_attn_qk_int8_pv_f16(
        _p(query), _p(key), _p(v_image), _p(output), _p(lse) if return_lse else None, _p(query_scale), _p(key_scale),
        _p(value_mean), B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
        is_causal, qk_quant_gran, q_warp, float(sm_scale_log2), pv_accum, code,
        torch._C._cuda_getCurrentRawStream(output.device.index), _cabi.attr_arg(attr))
    attn_check(rc, "sage_attn_qk_int8_pv_f16", attr, query.device)
    return lse


qk_int8_sv_f16_attn = torch.library.custom_op("sageattention_gfx950::qk_int8_sv_f16_attn", qk_int8_sv_f16_attn_impl,
                                              mutates_args=("output",), device_types="cuda")


@qk_int8_sv_f16_attn.register_fake
def _(query, key, v_image, output, query_scale, key_scale, value_mean, tensor_layout, is_causal, qk_quant_gran, q_warp,
      sm_scale_log2, pv_accum, return_lse):
    return _lse_alloc(query, tensor_layout, return_lse)


# ------------------------------------------------------------------------------------------------ whole-call op
# Under torch.compile the public entry points (core.sageattn*, dense) dispatch to ONE opaque op that runs the eager
# pipeline -- K mean, INT8 / FP8 pre-pass kernels, fused attention -- so a compiled model has no graph break around the
# attention call, with any backend (the pre-pass kernels go through ctypes, which dynamo cannot trace).
def sageattn_call_impl(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, api: str, tensor_layout: str, is_causal: bool,
                       qk_quant_gran: str, sm_scale: Optional[float], pv_accum_dtype: str, smooth_k: bool, smooth_v: bool,
                       return_lse: bool) -> tuple[torch.Tensor, torch.Tensor]:
    from . import core
    fn = {"fp8": core.sageattn_qk_int8_pv_fp8_cuda, "fp16": core.sageattn_qk_int8_pv_fp16_cuda,
          "sm90": core.sageattn_qk_int8_pv_fp8_cuda_sm90}[api]
    kw = dict(tensor_layout=tensor_layout, is_causal=is_causal, qk_quant_gran=qk_quant_gran, sm_scale=sm_scale,
              pv_accum_dtype=pv_accum_dtype, smooth_k=smooth_k, return_lse=return_lse)
    if api != "sm90":
        kw["smooth_v"] = smooth_v
    out = fn(q, k, v, **kw)
    o, lse = out if return_lse else (out, torch.empty((0,), dtype=torch.float32, device=q.device))
    return o.contiguous(), lse


sageattn_call = torch.library.custom_op("sageattention_gfx950::sageattn_call", sageattn_call_impl, mutates_args=(),
                                        device_types="cuda")


@sageattn_call.register_fake
def _(q, k, v, api, tensor_layout, is_causal, qk_quant_gran, sm_scale, pv_accum_dtype, smooth_k, smooth_v, return_lse):
    o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    if return_lse:
        B, H, L = (q.shape[0], q.shape[1], q.shape[2]) if tensor_layout == "HND" else (q.shape[0], q.shape[2], q.shape[1])
        return o, torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    return o, torch.empty((0,), dtype=torch.float32, device=q.device)
