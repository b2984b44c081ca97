"""Public API of sageattention_amd -- the drop-in for ``sageattention/core.py`` on MI355X.

Same six public names, same signatures, same kwargs and error behaviour as the reference
(``/root/reference/sageattention/core.py``: ``sageattn`` :79, ``sageattn_qk_int8_pv_fp16_triton``
:160, ``sageattn_varlen`` :334, ``sageattn_qk_int8_pv_fp16_cuda`` :451,
``sageattn_qk_int8_pv_fp8_cuda`` :636, ``sageattn_qk_int8_pv_fp8_cuda_sm90`` :829), so
``F.scaled_dot_product_attention = sageattn`` in the CogVideoX / Hunyuan / Wan examples keeps
working.  Host code here is plumbing only (padding, K mean, allocation, LSE fix-up -- exactly
what the reference does in Python); every tensor-sized computation, including the K-smoothing
mean, runs in the hand-written HIP kernels behind ``libsage_gfx950.so``.  There is no fallback path: CPU
tensors or a missing library raise.
"""
from __future__ import annotations

import os

import ctypes
import warnings
from typing import Any, Optional

import torch
import torch.nn.functional as F

from . import _cabi, ops
from .quant import (LOG2E, _aligned, _cu_blocks, _dims, _p, _quant, _squeeze_km, _stream, channel_mean, channel_mean_packed, per_block_int8, per_block_int8_varlen,
                    per_channel_fp8, prep_v_fp16, prep_v_fp16_varlen, prepass_fused_ok, prepass_kv_fp8, prepass_kv_varlen,
                    prepass_varlen_fused_ok, sub_mean, varlen_plan)

_SUPPORTED_ARCH_PREFIX = "gfx950"
_FUSE_Q16_DEFAULT = os.environ.get("SAGE_FUSE_Q16", "1") != "0"      # debugging switch


def get_gcn_arch(device: torch.device) -> str:
    """ROCm analogue of the reference's ``get_cuda_arch_versions`` (core.py:71-76)."""
    return torch.cuda.get_device_properties(device).gcnArchName.split(":")[0]


def _check_inputs(q, k, v):
    dtype = q.dtype
    assert q.is_cuda, "Input tensors must be on cuda."
    assert dtype in [torch.float16, torch.bfloat16], "Input tensors must be in dtype of torch.float16 or torch.bfloat16"
    assert q.device == k.device == v.device, "All tensors must be on the same device."
    assert q.dtype == k.dtype == v.dtype, "All tensors must have the same dtype."


def _pad_head_dim(q, k, v):
    """core.py:260-271: zero-pad head_dim to 64 / 128; > 128 is rejected."""
    head_dim_og = q.size(-1)
    if head_dim_og < 64:
        q, k, v = (F.pad(t, (0, 64 - head_dim_og)) for t in (q, k, v))
    elif 64 < head_dim_og < 128:
        q, k, v = (F.pad(t, (0, 128 - head_dim_og)) for t in (q, k, v))
    elif head_dim_og > 128:
        raise ValueError(f"Unsupported head_dim: {head_dim_og}")
    return q, k, v, head_dim_og


def _lse_correction(q, km, tensor_layout):
    """core.py:283-293: q . km^T per query row (fp32), km broadcast over the GQA group."""
    nh_dim = 2 if tensor_layout == "NHD" else 1
    g = q.size(nh_dim) // km.size(nh_dim)
    km_b = torch.repeat_interleave(km, g, dim=nh_dim) if g > 1 else km
    if tensor_layout == "NHD":
        return torch.matmul(q.transpose(1, 2), km_b.transpose(1, 2).transpose(2, 3)).squeeze(-1).to(torch.float32)
    return torch.matmul(q, km_b.transpose(2, 3)).squeeze(-1).to(torch.float32)


def _smooth_k(q, k, tensor_layout, smooth_k, return_lse):
    """core.py:279-295: km = mean of k over the sequence (keepdim), and q.km^T for the LSE fix-up."""
    if not smooth_k:
        return None, None
    seq_dim = 1 if tensor_layout == "NHD" else 2
    km = channel_mean(k, tensor_layout).unsqueeze(seq_dim)     # k.mean(dim=seq_dim, keepdim=True) as a HIP reduction
    return km, (_lse_correction(q, km, tensor_layout) if return_lse else None)


def _attn_dense(fp8, q_int8, k_int8, v_image, v_scale, q_scale, k_scale, out_dtype, tensor_layout, is_causal,
                gran, q_warp, sm_scale_log2, two_level, return_lse, v_mean=None, folded_scores=False):
    """Allocate ``o`` and launch the fused kernel through the registered custom op (-> C ABI)."""
    B, Hq, Lq, D, _, _, _ = _dims(q_int8, tensor_layout)
    Hkv = _dims(k_int8, tensor_layout)[1]
    assert Hq % Hkv == 0, "num_qo_heads must be divisible by num_kv_heads"
    o = torch.empty(q_int8.shape, dtype=out_dtype, device=q_int8.device)
    layout = 0 if tensor_layout == "NHD" else 1            # the reference's encoding (core.py:556)
    # two_level: True / False, or "triton" = the Triton kernel form of the FP16-PV entry point (_cabi.PV_ACCUM_TRITON)
    accum = _cabi.PV_ACCUM_TRITON if two_level == "triton" else (_cabi.PV_ACCUM_TWO_LEVEL if two_level else _cabi.PV_ACCUM_SINGLE)
    # Under torch.compile the registered custom ops are traced; in eager mode their implementations are
    # called directly (same code, minus ~15 us of dispatcher overhead per call).
    compiling = torch.compiler.is_compiling()
    f8 = ops.qk_int8_sv_f8_attn if compiling else ops.qk_int8_sv_f8_attn_impl
    f16 = ops.qk_int8_sv_f16_attn if compiling else ops.qk_int8_sv_f16_attn_impl
    if fp8:
        lse = f8(q_int8, k_int8, v_image, o, q_scale, k_scale, v_scale, v_mean, layout, int(is_causal),
                 gran, q_warp, float(sm_scale_log2), accum, int(return_lse), bool(folded_scores))
    else:
        lse = f16(q_int8, k_int8, v_image, o, q_scale, k_scale, v_mean, layout, int(is_causal),
                  gran, q_warp, float(sm_scale_log2), accum, int(return_lse))
    return o, (lse if return_lse else None)


def _v_rows_ok(v, tensor_layout: str) -> bool:
    """Whether an fp16 V tensor can be read in place by the FP16-PV kernels (``sage_attn_fused_q*_pv_f16_vrows``): 16-byte rows."""
    if v.dtype != torch.float16 or v.stride(-1) != 1 or v.data_ptr() % 16 != 0:
        return False
    _, _, L, D, sb, sh, sl = _dims(v, tensor_layout)
    return sb % 8 == 0 and sh % 8 == 0 and sl % 8 == 0 and sl >= D and ((L - 1) * sl + D) * 2 < 2 ** 31


_V_IN_PLACE = {"0": False, "1": True}.get(os.environ.get("SAGE_V_IN_PLACE", ""))      # 0 / 1: force the route for every eligible call (A/B, debugging)


def _v_rows_wanted(q, k, v, tensor_layout: str, is_causal: bool, override) -> bool:
    """Whether an FP16-PV call on fp16 inputs reads V's rows in place instead of building the tile image.  Same bits either way.  Measured
    (profiles/r6_run_e_vrows_ab.txt, B2 H32 D128): the K-only pre-pass is 24-100 us shorter than K + V image (46 vs 72 us at N = 4096, 116 vs
    214 at 16384: 4 of its 7 bytes per element gone), the attention kernel 2.2-3.7 % slower (twice the LDS read instructions for the V operand:
    64-bit transposing reads); whole call + 6.5 % at C2 (N = 4096 causal), + 7-8 % at N = 2048, + 1.6-3 % at N = 4096 non-causal, 0 at N = 16384
    causal.  The saving grows with Lk per kv-head, the loss with the (query, key) pairs per query head: rows in place up to
    (Hq / Hkv) * Lq * (1/2 if causal) = 6144, the image beyond."""
    if not _v_rows_ok(v, tensor_layout):
        return False
    if override is None:
        override = _V_IN_PLACE
    if override is not None:
        return bool(override)
    _, Hq, Lq = _dims(q, tensor_layout)[:3]
    Hkv = _dims(k, tensor_layout)[1]
    return (Hq // Hkv) * Lq * (0.5 if is_causal else 1.0) <= 6144


@torch.compiler.disable
def _attn_fused_q(q, k_int8, v_image, v_scale, k_scale, tensor_layout, is_causal, sm_scale_log2, return_lse, v_mean=None, folded_scores=False,
                  v_rows=False):
    """FP8-PV two-level attention with the per-thread Q quantisation done in the kernel prologue
    (``sage_attn_fused_q_pv_f8``): bit-identical to ``per_thread_int8`` + the attention op, one launch and
    3 B/element of HBM traffic less.  ``v_rows`` (FP16 PV): ``v_image`` is the fp16 V tensor itself, read in place."""
    B, Hq, Lq, D, q_sb, q_sh, q_sl = _dims(q, tensor_layout)
    _, Hkv, Lk, _, k_sb, k_sh, k_sl = _dims(k_int8, tensor_layout)
    assert Hq % Hkv == 0, "num_qo_heads must be divisible by num_kv_heads"
    o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    _, _, _, _, o_sb, o_sh, o_sl = _dims(o, tensor_layout)
    lse = torch.empty((B, Hq, Lq), dtype=torch.float32, device=q.device) if return_lse else None
    code = _cabi.DTYPE_F16 if q.dtype == torch.float16 else _cabi.DTYPE_BF16
    # (a large non-causal call: persistent launch; FP8 PV: the score form)
    attr = ops.attn_attr(q.device, is_causal, B * Hq * ((Lq + 127) // 128), folded_scores and v_scale is not None)
    if v_rows:                     # FP16 PV on fp16 inputs: V rows in place, no tile image (sage_attn_fused_q_pv_f16_vrows)
        assert v_scale is None and v_mean is None
        _, _, _, _, v_sb, v_sh, v_sl = _dims(v_image, tensor_layout)
        rc = _cabi.load().sage_attn_fused_q_pv_f16_vrows(
            _p(q), _p(k_int8), _p(v_image), _p(o), _p(lse), _p(k_scale),
            B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl,
            int(is_causal), float(sm_scale_log2), code, _stream(q), _cabi.attr_arg(attr))
        ops.attn_check(rc, "sage_attn_fused_q_pv_f16_vrows", attr, q.device)
        return o, lse
    if v_scale is None:            # FP16 PV (v_image from prep_v_fp16), straight FP32 accumulation
        rc = _cabi.load().sage_attn_fused_q_pv_f16(
            _p(q), _p(k_int8), _p(v_image), _p(o), _p(lse), _p(k_scale), _p(v_mean),
            B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
            int(is_causal), float(sm_scale_log2), code, code, _stream(q), _cabi.attr_arg(attr))
        ops.attn_check(rc, "sage_attn_fused_q_pv_f16", attr, q.device)
        return o, lse
    rc = _cabi.load().sage_attn_fused_q_pv_f8(
        _p(q), _p(k_int8), _p(v_image), _p(o), _p(lse), _p(k_scale), _p(v_scale), _p(v_mean),
        B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
        int(is_causal), float(sm_scale_log2), code, code, _stream(q), _cabi.attr_arg(attr))
    ops.attn_check(rc, "sage_attn_fused_q_pv_f8", attr, q.device)
    return o, lse


@torch.compiler.disable
def _attn_fused_qblock(q, k_int8, v_image, k_scale, tensor_layout, is_causal, q_premul, return_lse, v_rows=False):
    """The Triton-named API's attention with the per-block Q quantisation in the kernel prologue
    (``sage_attn_fused_qblock_pv_f16``): bit-identical to ``per_block_int8`` (q half) + the attention op.  ``v_rows``: ``v_image`` is the
    fp16 V tensor itself, read in place (``sage_attn_fused_qblock_pv_f16_vrows``)."""
    q = _aligned(q, 8)
    B, Hq, Lq, D, q_sb, q_sh, q_sl = _dims(q, tensor_layout)
    _, Hkv, Lk, _, k_sb, k_sh, k_sl = _dims(k_int8, tensor_layout)
    assert Hq % Hkv == 0, "num_qo_heads must be divisible by num_kv_heads"
    o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    _, _, _, _, o_sb, o_sh, o_sl = _dims(o, tensor_layout)
    lse = torch.empty((B, Hq, Lq), dtype=torch.float32, device=q.device) if return_lse else None
    code = _cabi.DTYPE_F16 if q.dtype == torch.float16 else _cabi.DTYPE_BF16
    attr = ops.attn_attr(q.device, is_causal, B * Hq * ((Lq + 127) // 128))
    if v_rows:
        _, _, _, _, v_sb, v_sh, v_sl = _dims(v_image, tensor_layout)
        rc = _cabi.load().sage_attn_fused_qblock_pv_f16_vrows(
            _p(q), _p(k_int8), _p(v_image), _p(o), _p(lse), _p(k_scale),
            B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl,
            int(is_causal), float(q_premul), code, _stream(q), _cabi.attr_arg(attr))
        ops.attn_check(rc, "sage_attn_fused_qblock_pv_f16_vrows", attr, q.device)
        return o, lse
    rc = _cabi.load().sage_attn_fused_qblock_pv_f16(
        _p(q), _p(k_int8), _p(v_image), _p(o), _p(lse), _p(k_scale),
        B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl,
        int(is_causal), float(q_premul), code, code, _stream(q), _cabi.attr_arg(attr))
    ops.attn_check(rc, "sage_attn_fused_qblock_pv_f16", attr, q.device)
    return o, lse


def _split_kv_plan(B: int, Hq: int, Lq: int, Lk: int, is_causal: bool, override, auto_default: bool = True) -> int:
    """Number of key-range chunks S for a call whose grid would not fill the chip (0 = no split).  The reference kernels
    parallelise over (batch, head, 128-row q block) only, so few query rows against a long key range (cross-attention,
    decode-like shapes) leave most of the 256 CUs idle.  Such a call runs as
    S chunks of Lk/S keys folded into the kv-head dimension and one merge by log-sum-exp.  Chunks are whole numbers of
    64-key tiles (the quantisation groups and the V image tiles are unchanged by the fold), so only S | Lk/64 is considered.
    ``override``: 0 = never, an integer S >= 2 = that split, "auto" = the plan below, None = ``auto_default``.  The FP16-PV entry point
    plans by default (a split result meets the UNSPLIT oracle at the kernel tolerance: P is rounded to fp16).  The FP8-PV entry points do
    NOT (round 5): a split changes which running maximum every P is rounded to e4m3 against -- measured rel-RMS up to 2.8e-2 against the
    unsplit oracle, beyond the 1e-2 that a schedule variant may differ from the exact schedule by and still be a DEFAULT FP8 route
     -- so there it is opt-in (``split_kv="auto"`` or S), held to 2e-3 against the schedule-matched oracle as before."""
    if override is None:
        override = "auto" if auto_default else 0
    if override == 0:
        return 0
    if override != "auto":
        if isinstance(override, bool) or not isinstance(override, int):
            raise ValueError(f"split_kv={override!r}: 0, an integer >= 2 or 'auto'")
        if Lk % 64 != 0 or override < 2 or (Lk // 64) % override != 0:
            raise ValueError(f"split_kv={override} must be >= 2 and divide the number of whole 64-key tiles (kv_len {Lk})")
        return override
    if Lk % 64 != 0:
        return 0
    ntk = Lk // 64
    n_wg = B * Hq * ((Lq + 127) // 128)
    if is_causal:
        # Causal calls are split only on request (split_kv=S; the mask then runs in global key coordinates).  Measured for the
        # case it could help -- one partial wave of workgroups, B=1 H=8 N=8192: 179 us unsplit, 199 us with S=4 -- the partial
        # outputs' extra pass through HBM and the merge cost more than the better balance returns (profiles/r2_run_z_shape_probe.txt).
        return 0
    else:
        if n_wg > 128 or ntk < 64:                     # the grid already covers half the chip, or the key range is short
            return 0
        target = max(2, min(ntk // 32, -(-768 // n_wg)))   # chunks of >= 32 tiles, about three workgroups per CU
    S = max(d for d in range(1, target + 1) if ntk % d == 0)
    return S if S >= 2 else 0


@torch.compiler.disable
def _attn_fused_q_split(q, k_int8, v_image, v_scale, k_scale, tensor_layout, is_causal, sm_scale_log2, S, return_lse, v_mean=None,
                        folded_scores=False):
    """Split-KV route of the fused-Q FP8 attention: the key range in S chunks folded into the kv-head dimension (zero-copy
    views of the INT8 K, its scales and the V image; Q is read in place by every chunk), partial outputs in fp16 +
    log2-domain log-sum-exps, one ``sage_merge_split`` pass."""
    B, Hq, Lq, D, q_sb, q_sh, q_sl = _dims(q, tensor_layout)
    _, Hkv, Lk, _, _, _, _ = _dims(k_int8, tensor_layout)
    group, Lc = Hq // Hkv, Lk // S
    k_store = k_int8 if tensor_layout == "HND" else k_int8.permute(0, 2, 1, 3)      # head-major storage (quant._quant)
    assert k_store.is_contiguous() and v_image.is_contiguous() and k_scale.is_contiguous()
    k_f = k_store.view(B, Hkv * S, Lc, D)
    ks_f = k_scale.view(B, Hkv * S, -1)
    vs_f = None if v_scale is None else v_scale.repeat_interleave(S, dim=1)      # None: FP16 PV (fp16 V image)
    vm_f = None if v_mean is None else v_mean.repeat_interleave(S, dim=1)
    o_part = torch.empty((B, Hq * S, Lq, D), dtype=torch.float16, device=q.device)
    lse_part = torch.empty((B, Hq * S, Lq), dtype=torch.float32, device=q.device)
    _, _, _, _, k_sb, k_sh, k_sl = _dims(k_f, "HND")
    _, _, _, _, p_sb, p_sh, p_sl = _dims(o_part, "HND")
    code = _cabi.DTYPE_F16 if q.dtype == torch.float16 else _cabi.DTYPE_BF16
    lib = _cabi.load()
    if vs_f is None:
        rc = lib.sage_attn_fused_q_pv_f16_split(
            _p(q), _p(k_f), _p(v_image), _p(o_part), _p(lse_part), _p(ks_f), _p(vm_f),
            B, Hq, Hkv, S, Lq, Lc, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, p_sb, p_sh, p_sl,
            int(is_causal), float(sm_scale_log2), code, _cabi.DTYPE_F16, _stream(q), None)
        _cabi.check(rc, "sage_attn_fused_q_pv_f16_split")
    else:
        rc = lib.sage_attn_fused_q_pv_f8_split(
            _p(q), _p(k_f), _p(v_image), _p(o_part), _p(lse_part), _p(ks_f), _p(vs_f), _p(vm_f),
            B, Hq, Hkv, S, Lq, Lc, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, p_sb, p_sh, p_sl,
            int(is_causal), float(sm_scale_log2), code, _cabi.DTYPE_F16, _stream(q),
            _cabi.attr_arg(_cabi.launch_attr(folded_scores=bool(folded_scores))))
        _cabi.check(rc, "sage_attn_fused_q_pv_f8_split")
    o = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    _, _, _, _, o_sb, o_sh, o_sl = _dims(o, tensor_layout)
    lse = torch.empty((B, Hq, Lq), dtype=torch.float32, device=q.device) if return_lse else None
    rc = lib.sage_merge_split(_p(o_part), _p(lse_part), None, None, _p(o), _p(lse), B, S, Hq, group, Lq, D,
                              o_sb, o_sh, o_sl, code, _stream(q))
    _cabi.check(rc, "sage_merge_split")
    return o, lse


@torch.compiler.disable
def _attn_masked(q_int8, k_int8, v_image, q_scale, k_scale, attn_mask, out_dtype, tensor_layout, return_lse):
    """Triton-named API with ``attn_mask`` (core.py:313-324): the mask is broadcast to
    ``[B, Hq, Lq, Lk]`` by ``expand`` (zero strides, no copy) and read in place by the kernel."""
    B, Hq, Lq, D, q_sb, q_sh, q_sl = _dims(q_int8, tensor_layout)
    _, Hkv, Lk, _, k_sb, k_sh, k_sl = _dims(k_int8, tensor_layout)
    assert Hq % Hkv == 0, "num_qo_heads must be divisible by num_kv_heads"
    target_shape = (B, Hq, Lq, Lk)
    try:
        attn_mask = attn_mask.expand(target_shape)
    except Exception:
        raise AssertionError(f"attn_mask shape {attn_mask.shape} cannot be broadcast to {target_shape}")
    kind = _cabi.MASK_BOOL if attn_mask.dtype == torch.bool else (_cabi.MASK_F16 if attn_mask.dtype == torch.float16 else _cabi.MASK_BF16)
    o = torch.empty(q_int8.shape, dtype=out_dtype, device=q_int8.device)
    _, _, _, _, o_sb, o_sh, o_sl = _dims(o, tensor_layout)
    lse = torch.empty((B, Hq, Lq), dtype=torch.float32, device=o.device) if return_lse else None
    code = _cabi.DTYPE_F16 if out_dtype == torch.float16 else _cabi.DTYPE_BF16
    rc = _cabi.load().sage_attn_qk_int8_pv_f16_masked(
        _p(q_int8), _p(k_int8), _p(v_image), _p(o), _p(lse), _p(q_scale), _p(k_scale), _p(attn_mask), kind,
        attn_mask.stride(0), attn_mask.stride(1), attn_mask.stride(2), attn_mask.stride(3),
        B, Hq, Hkv, Lq, Lk, D, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, o_sb, o_sh, o_sl, 1.0, code, _stream(o), None)
    _cabi.check(rc, "sage_attn_qk_int8_pv_f16_masked")
    return o, lse


def _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale):
    o = o[..., :head_dim_og]
    if return_lse:   # core.py:328-329: kernel LSE is in log2 units
        return o, lse / 1.44269504 + lse_correction * sm_scale if smooth_k else lse / 1.44269504
    return o


# ------------------------------------------------------------------------------------------------
def sageattn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, tensor_layout: str = "HND", is_causal: bool = False,
             sm_scale: Optional[float] = None, return_lse: bool = False, **kwargs: Any):
    """Select the implementation for the device, as the reference does per compute capability
    (core.py:143-157).  On gfx950 that is INT8 QK^T + FP8 PV with two-level FP32 accumulation
    (the reference's sm90 choice, ``pv_accum_dtype="fp32+fp32"``).  Extra SDPA-style kwargs
    (``attn_mask=``, ``dropout_p=``, ``scale=`` ...) are accepted and ignored exactly as the
    reference ignores them."""
    if torch.compiler.is_compiling():      # the device query is not traceable; the opaque op checks the device when it runs
        return sageattn_qk_int8_pv_fp8_cuda(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                                            return_lse=return_lse, pv_accum_dtype="fp32+fp32")
    arch = get_gcn_arch(q.device) if q.is_cuda else "cpu"
    if arch.startswith(_SUPPORTED_ARCH_PREFIX):
        return sageattn_qk_int8_pv_fp8_cuda(q, k, v, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=sm_scale,
                                            return_lse=return_lse, pv_accum_dtype="fp32+fp32", split_kv=kwargs.get("split_kv"),
                                            fused_prepass=kwargs.get("fused_prepass"), fp8_scores=kwargs.get("fp8_scores"))
    raise ValueError(f"Unsupported architecture: {arch} (sageattention_amd targets gfx950 / MI355X only)")


def sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout: str = "HND", quantization_backend: str = "triton",
                                    is_causal: bool = False, attn_mask: Optional[torch.Tensor] = None,
                                    sm_scale: Optional[float] = None, smooth_k: bool = True, return_lse: bool = False,
                                    **kwargs: Any):
    """Per-block INT8 Q/K (sm_scale*log2e folded into Q), FP16 PV, per-tile product added to an
    FP32 buffer (reference core.py:160-331; the name is kept for drop-in -- there is no Triton
    here, the same HIP kernel family runs it)."""
    dtype = q.dtype
    _check_inputs(q, k, v)
    if attn_mask is not None:
        assert attn_mask.dtype == torch.bool or attn_mask.dtype == q.dtype, "attn_mask must be of dtype bool or the same dtype as q."
        assert attn_mask.device == q.device, "All tensors must be on the same device."
    if quantization_backend not in ("triton", "cuda"):
        raise ValueError(f"Unsupported quantization backend: {quantization_backend}")
    torch.cuda.set_device(v.device)
    q, k, v, head_dim_og = _pad_head_dim(q, k, v)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "Last dim of qkv must be contiguous."
    # the Q half of the per-block quantiser runs in the attention kernel's prologue (same bits, no INT8 copy of Q in HBM) unless the
    # CUDA rounding convention or a mask is asked for; fp16 inputs on that route need no V pass at all: the kernel reads V's rows in place
    # (the reference's `v.to(torch.float16)`, core.py:297-298, is the identity for them) -- same bits as the image route
    fuse_q = quantization_backend == "triton" and attn_mask is None and kwargs.get("fuse_q_quant", True)
    v_rows = fuse_q and _v_rows_wanted(q, k, v, tensor_layout, is_causal, kwargs.get("v_in_place"))
    # K mean + INT8 K (Triton rounding) + the fp16 V image as ONE launch that reads K and V once (sage_prepass_kv), when it is the faster route
    k_done = v_image = None
    if quantization_backend == "triton" and k.shape == v.shape and _fused_prepass_wanted(k, tensor_layout, kwargs.get("fused_prepass")):
        km_s, k8, ks, v_image, _, _ = prepass_kv_fp8(k, None if v_rows else v, tensor_layout, smooth_k=smooth_k, qk_quant_gran="per_block_triton",
                                                     v_fp16=True)
        k_done = (k8, ks)
        km, lse_correction = None, None
        if smooth_k:
            km = km_s.unsqueeze(1 if tensor_layout == "NHD" else 2)
            lse_correction = _lse_correction(q, km, tensor_layout) if return_lse else None
    else:
        km, lse_correction = _smooth_k(q, k, tensor_layout, smooth_k, return_lse)
    if sm_scale is None:
        sm_scale = 1.0 / (head_dim_og ** 0.5)
    if is_causal:
        assert q.size(1 if tensor_layout == "NHD" else 2) == k.size(1 if tensor_layout == "NHD" else 2), \
            "qo_len and kv_len must be equal for causal attention"
    if is_causal:
        assert attn_mask is None, "Mask should be None for causal attention."        # core.py:310
    q_int8, q_scale, k_int8, k_scale = per_block_int8(None if fuse_q else q, k, km=km, sm_scale=sm_scale, tensor_layout=tensor_layout,
                                                      quantization_backend=quantization_backend, k_done=k_done)
    if v_image is None and not v_rows:
        v_image = prep_v_fp16(v, tensor_layout)
    if fuse_q:
        o, lse = _attn_fused_qblock(q, k_int8, v if v_rows else v_image, k_scale, tensor_layout, is_causal, sm_scale * LOG2E, return_lse, v_rows=v_rows)
    elif attn_mask is not None:
        o, lse = _attn_masked(q_int8, k_int8, v_image, q_scale, k_scale, attn_mask, dtype, tensor_layout, return_lse)
    else:
        o, lse = _attn_dense(False, q_int8, k_int8, v_image, None, q_scale, k_scale, dtype, tensor_layout, is_causal,
                             _cabi.GRAN_PER_BLOCK, 128, 1.0, "triton", return_lse)
    return _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale)


class _VarlenState:
    """Operands of the attention launch of one ``sageattn_varlen`` call (what its pre-pass produces)."""
    __slots__ = ("q", "q_int8", "q_scale", "k_int8", "k_scale", "v_image", "cu_q", "cu_k", "cu_qs", "cu_ks", "order", "plan", "fuse_q",
                 "max_seqlen_q", "is_causal", "q_premul", "dtype", "head_dim_og")


@torch.compiler.disable
def _varlen_prepare(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, is_causal, sm_scale, smooth_k, kwargs) -> _VarlenState:
    """Everything of ``sageattn_varlen`` in front of the attention launch (core.py:427-444): the index arrays (one launch, no host
    synchronisation), ``km`` over all packed tokens, INT8 K, the fp16 V image -- one launch that reads K and V once where the head barrier
    reaches (``prepass_kv_varlen``), else the kernel sequence with the same bits."""
    st = _VarlenState()
    st.dtype = q.dtype
    _check_inputs(q, k, v)
    torch.cuda.set_device(v.device)
    q, k, v, st.head_dim_og = _pad_head_dim(q, k, v)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "Last dim of qkv must be contiguous."
    assert cu_seqlens_q.is_contiguous() and cu_seqlens_k.is_contiguous(), "cu_seqlens_q and cu_seqlens_k must be contiguous."
    Hq, Hkv, D = q.shape[1], k.shape[1], q.shape[2]
    assert Hq % Hkv == 0, "num_qo_heads must be divisible by num_kv_heads"
    if sm_scale is None:
        sm_scale = 1.0 / (st.head_dim_og ** 0.5)
    st.fuse_q = fuse_q = kwargs.get("fuse_q_quant", True)      # the Q half of the quantiser in the attention kernel's prologue (same bits)
    st.cu_q = cu_q = cu_seqlens_q.to(torch.int32).contiguous()
    st.cu_k = cu_k = cu_seqlens_k.to(torch.int32).contiguous()
    nseq = cu_q.shape[0] - 1
    # block-count prefix sums, the attention launch's work list and the pre-pass's slab map from one small launch (None: more sequences than
    # it takes -- then torch prefix sums, an on-device argsort for the order and the kernel sequence)
    plan = varlen_plan(cu_q, cu_k, want_q_blocks=not fuse_q, total_q=q.shape[0], total_k=k.shape[0], is_causal=is_causal, Hq=Hq, Hkv=Hkv,
                       head_dim=D, pv_fp8=False) if kwargs.get("varlen_plan", True) else None
    st.plan = plan if (plan is not None and kwargs.get("work_list", True)) else None
    fused = kwargs.get("fused_prepass")
    if fused is None:
        fused = os.environ.get("SAGE_PREPASS", "") not in ("seq", "sequence", "0")
    fused = bool(fused) and k.shape == v.shape and prepass_varlen_fused_ok(k, plan, max_seqlen_k, smooth_k)
    if fused:
        _, st.k_int8, st.k_scale, st.v_image = prepass_kv_varlen(k, v, cu_k, plan, max_seqlen_k, smooth_k=smooth_k)
        st.cu_ks = plan.cu_ks
        st.q_int8 = st.q_scale = st.cu_qs = None
        if not fuse_q:
            st.q_int8, st.q_scale, _, _, st.cu_qs, _ = per_block_int8_varlen(q, None, cu_q, cu_k, max_seqlen_q, max_seqlen_k, sm_scale=sm_scale,
                                                                            cu_qs=plan.cu_qs)
    else:
        km = channel_mean_packed(k, cu_k, plan) if smooth_k else None   # mean over ALL packed tokens, as core.py:432-434
        # (prefix arrays from the plan, or from torch ops on the device: either way the scale tensors are allocated at their host-known
        #  bounds and nothing synchronises -- the reference's `.item()` pair is gone on every route)
        st.q_int8, st.q_scale, st.k_int8, st.k_scale, st.cu_qs, st.cu_ks = per_block_int8_varlen(
            None if fuse_q else q, k, cu_q, cu_k, max_seqlen_q, max_seqlen_k, km=km, sm_scale=sm_scale,
            cu_ks=plan.cu_ks if plan is not None else _cu_blocks(cu_k, 64),
            cu_qs=None if fuse_q else (plan.cu_qs if plan is not None else _cu_blocks(cu_q, 128)))
        st.v_image = prep_v_fp16_varlen(v, cu_k, st.cu_ks, max_seqlen_k, ntiles=(k.shape[0] + 63) // 64 + nseq)
    # without a work list: schedule the longest sequences first (on the device, no sync); results do not depend on the order
    st.order = plan.order if plan is not None else torch.argsort(cu_q[1:] - cu_q[:-1], descending=True).to(torch.int32)
    st.q = _aligned(q, 8) if fuse_q else None
    st.max_seqlen_q, st.is_causal, st.q_premul = int(max_seqlen_q), bool(is_causal), float(sm_scale * LOG2E)
    return st


@torch.compiler.disable
def _varlen_attend(st: _VarlenState) -> torch.Tensor:
    """The attention launch of ``sageattn_varlen`` (attn_qk_int8_block_varlen.py:123, _causal_varlen.py:125)."""
    q = st.q if st.fuse_q else st.q_int8
    Hq, D = q.shape[1], q.shape[2]
    Hkv = st.k_int8.shape[1]
    o = torch.empty(q.shape, dtype=st.dtype, device=q.device)
    code = _cabi.DTYPE_F16 if st.dtype == torch.float16 else _cabi.DTYPE_BF16
    plan = st.plan
    items, hdr, bound = (plan.items, plan.hdr, plan.items_bound) if plan is not None else (None, None, 0)
    nseq = st.cu_q.shape[0] - 1
    # (a large call over the work list runs as a persistent launch -- causal too, on the fused-Q route whose causal kernels carry the ticket
    #  loop; items = 128-row blocks of the packed rows, at least)
    attr = ops.attn_attr(q.device, st.is_causal, Hq * (q.shape[0] // 128), packed=st.fuse_q) if plan is not None else None
    if st.fuse_q:
        rc = _cabi.load().sage_attn_fused_qblock_pv_f16_varlen(
            _p(q), _p(st.k_int8), _p(st.v_image), _p(o), _p(st.k_scale), _p(st.cu_q), _p(st.cu_k), _p(st.cu_ks), _p(st.order),
            _p(items), _p(hdr), bound, nseq, st.max_seqlen_q, Hq, Hkv, D, q.stride(0), q.stride(1), st.k_int8.stride(0), st.k_int8.stride(1),
            o.stride(0), o.stride(1), int(st.is_causal), st.q_premul, code, code, _stream(o), _cabi.attr_arg(attr))
        ops.attn_check(rc, "sage_attn_fused_qblock_pv_f16_varlen", attr, q.device)
    else:
        rc = _cabi.load().sage_attn_qk_int8_pv_f16_varlen(
            _p(q), _p(st.k_int8), _p(st.v_image), _p(o), _p(st.q_scale), _p(st.k_scale), _p(st.cu_q), _p(st.cu_k), _p(st.cu_qs), _p(st.cu_ks),
            _p(st.order), _p(items), _p(hdr), bound, nseq, st.max_seqlen_q, Hq, Hkv, D, q.stride(0), q.stride(1),
            st.k_int8.stride(0), st.k_int8.stride(1), o.stride(0), o.stride(1), int(st.is_causal), 1.0, _cabi.PV_ACCUM_TRITON, code, _stream(o),
            _cabi.attr_arg(attr))
        ops.attn_check(rc, "sage_attn_qk_int8_pv_f16_varlen", attr, q.device)
    return o[..., :st.head_dim_og]


@torch.compiler.disable
def sageattn_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q: int, max_seqlen_k: int, is_causal: bool = False,
                    sm_scale: Optional[float] = None, smooth_k: bool = True, **kwargs: Any) -> torch.Tensor:
    """Variable-length batches, q/k/v packed as ``[sum L, H, D]`` (reference core.py:334-448).  Three launches and no host
    synchronisation on the default route: the index arrays (``sage_varlen_plan``), the K / V pre-pass (``sage_prepass_kv_varlen``) and
    the attention kernel over a device-built work list (every workgroup one existing query block, heaviest first) with the per-block Q
    quantisation in its prologue.  Route switches (same bits either way): ``fused_prepass=False`` the kernel sequence,
    ``work_list=False`` the unit order sized by ``max_seqlen_q``, ``fuse_q_quant=False`` a separate Q quantiser; ``varlen_plan=False`` builds
    the index arrays with torch ops as for more than 1024 sequences (the K mean is then summed over packed 512-token slabs instead of
    per-sequence ones: equal up to the last bit of an input-dtype rounding)."""
    return _varlen_attend(_varlen_prepare(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, is_causal, sm_scale, smooth_k,
                                          kwargs))


_ROUTE_KWARGS = ("split_kv", "fused_prepass", "fuse_q_quant", "fp8_scores", "v_in_place")


def _compiled_call(api, q, k, v, tensor_layout, is_causal, qk_quant_gran, sm_scale, pv_accum_dtype, smooth_k, smooth_v, return_lse,
                   kwargs=None):
    """torch.compile route of the dense CUDA-named entry points: one opaque custom op around the eager pipeline (ops.py).
    The op takes the default routes; a route override (``split_kv`` / ``fused_prepass`` / ``fuse_q_quant``) cannot travel through
    it, and a compiled run that silently took another route than the eager one would be a trap, so it is refused."""
    given = [n for n in _ROUTE_KWARGS if kwargs and kwargs.get(n) is not None]
    if given:
        raise ValueError(f"{', '.join(given)}: route overrides are not supported under torch.compile (the compiled op takes the default routes)")
    o, lse = ops.sageattn_call(q, k, v, api, tensor_layout, bool(is_causal), qk_quant_gran,
                               None if sm_scale is None else float(sm_scale), pv_accum_dtype, bool(smooth_k), bool(smooth_v),
                               bool(return_lse))
    return (o, lse) if return_lse else o


def _sm_log2(sm_scale: float) -> float:
    """sm_scale * log2(e) as the CUDA kernels form it: the Python double becomes a float kernel argument and is
    multiplied by the fp32 constant in fp32 (`sm_scale *= math::log2e`, qk_int_sv_f8_cuda_sm89.cuh:90, math.cuh:32).
    The product of two fp32 numbers is exact in a double, so one rounding to c_float is the fp32 product."""
    return ctypes.c_float(ctypes.c_float(sm_scale).value * ctypes.c_float(1.44269504088896340736).value).value


def _quant_q(q, qk_quant_gran, tensor_layout, warpq, sm_scale, blkk=64):
    """INT8 Q for the CUDA-named entry points.  Returns (q_int8, q_scale, gran code, q_warp, sm_scale_log2).  ``warpq``
    (32 / 16) and ``blkk`` (64 / 128) are the reference kernels' scale-group sizes (core.py:602-604, 964-970); the
    row -> group maps do not depend on the q block size, so BLKQ = 128 here also reproduces the sm90 kernels' BLKQ = 64 groups."""
    kflag = _cabi.GRAN_KBLK128 if blkk == 128 else 0
    if qk_quant_gran == "per_warp":            # quant.py:105-180 (q half)
        q_int8, q_scale = _quant(q, None, 128, warpq, _cabi.GRAN_PER_WARP, False, _cabi.QSTYLE_CUDA, 1.0, tensor_layout, 128 // warpq)
        return q_int8, q_scale, _cabi.GRAN_PER_WARP | kflag, warpq, _sm_log2(sm_scale)
    if qk_quant_gran == "per_thread":          # quant_per_thread.py:154-203 (q half)
        q_int8, q_scale = _quant(q, None, 128, warpq, _cabi.GRAN_PER_THREAD, False, _cabi.QSTYLE_TRITON_THREAD, 1.0, tensor_layout,
                                 (128 // warpq) * 8)
        return q_int8, q_scale, _cabi.GRAN_PER_THREAD | kflag, warpq, _sm_log2(sm_scale)
    if blkk != 64:
        raise ValueError("per_block scales are defined for 64-key groups only")
    # "per_block": gfx950 extension (the Triton path's granularity with the CUDA rounding); sm_scale * log2e folded into q
    q_int8, q_scale = _quant(q, None, 128, 128, _cabi.GRAN_PER_BLOCK, False, _cabi.QSTYLE_CUDA, sm_scale * LOG2E, tensor_layout, 1)
    return q_int8, q_scale, _cabi.GRAN_PER_BLOCK, 128, 1.0


def _fused_prepass_wanted(k, tensor_layout: str, override: Optional[bool]) -> bool:
    """Whether the K / V pre-pass runs as the one-launch, single-read kernel (``sage_prepass_kv``) or as the
    mean -> quantise -> statistics -> image sequence.  Same bits either way.  Measured on MI355X
    (profiles/r2_run_r3g_prepass_sweep.txt): the one launch wins from 512 keys up (129 vs 151 us at B=2 H=32 N=8192 D=128, 77 vs
    108 us at H=8 N=32768, 2x on launch-bound small calls) and loses only when very many heads of <= 256 keys leave its
    512-row slabs half empty (57 vs 47 us at B=64 H=16 N=256 D=64)."""
    if not prepass_fused_ok(k, tensor_layout):
        return False
    if override is not None:
        return bool(override)
    env = os.environ.get("SAGE_PREPASS", "")      # "seq" / "fused": force a route for every call (debugging, CU-masked streams)
    if env in ("seq", "sequence", "0"):
        return False
    if env in ("fused", "1"):
        return True
    B, H, L = _dims(k, tensor_layout)[:3]
    if L > 32768:
        # heads of 65 .. 128 slabs are within the barrier's reach (sage_prepass_max_seqlen = 65536) but every workgroup then waits for
        # 128 others: 325 vs 311 us at B=1 H=16 N=65536 (profiles/r3_run_f_prepass_64k.txt) -- the sequence unless the caller insists
        return False
    return L > 256 or B * H <= 256


def _prepass_kv(q, k, v, tensor_layout, qk_quant_gran, blkk, smooth_k, smooth_v, return_lse, fused: bool, v_fp8: bool = True,
                v_fp16: bool = False):
    """K mean + INT8 K (+ FP8 V image when ``v_fp8``).  Returns (lse_correction, km [B,H,D] | None, k_int8, k_scale, v_image,
    v_scale, vm); the K conventions are those of ``per_thread_int8`` / ``per_warp_int8`` / ``per_block_int8(cuda)``."""
    if fused:
        km_s, k_int8, k_scale, v_image, v_scale, vm = prepass_kv_fp8(k, v if (v_fp8 or v_fp16) else None, tensor_layout, smooth_k=smooth_k,
                                                                     smooth_v=smooth_v, BLKK=blkk, qk_quant_gran=qk_quant_gran, v_fp16=v_fp16)
        lse_correction = None
        if smooth_k and return_lse:
            lse_correction = _lse_correction(q, km_s.unsqueeze(1 if tensor_layout == "NHD" else 2), tensor_layout)
        return lse_correction, km_s, k_int8, k_scale, v_image, v_scale, vm
    km, lse_correction = _smooth_k(q, k, tensor_layout, smooth_k, return_lse)
    km_s = _squeeze_km(km, tensor_layout)
    if qk_quant_gran == "per_thread":
        k_int8, k_scale = _quant(k, km_s, blkk, blkk, _cabi.GRAN_PER_THREAD, True, _cabi.QSTYLE_TRITON_THREAD, 1.0, tensor_layout, 4)
    else:
        k_int8, k_scale = _quant(k, km_s, blkk, blkk, _cabi.GRAN_PER_BLOCK, True, _cabi.QSTYLE_CUDA, 1.0, tensor_layout, 1)
    v_image = v_scale = vm = None
    if v_fp8:
        v_image, v_scale, vm = per_channel_fp8(v, tensor_layout=tensor_layout, scale_max=448.0, smooth_v=smooth_v)
    return lse_correction, km_s, k_int8, k_scale, v_image, v_scale, vm


def sageattn_qk_int8_pv_fp16_cuda(q, k, v, tensor_layout: str = "HND", is_causal: bool = False,
                                  qk_quant_gran: str = "per_thread", sm_scale: Optional[float] = None,
                                  pv_accum_dtype: str = "fp32", smooth_k: bool = True, smooth_v: bool = False,
                                  return_lse: bool = False, **kwargs: Any):
    """INT8 QK^T + FP16 PV (reference core.py:451-633).  All three ``pv_accum_dtype`` values run the same arithmetic here: P.V is
    accumulated in FP32 (CDNA4's FP16 MFMA has no FP16 accumulator, so the reference's FP16 accumulator and its FP16 tile buffer
    have no counterpart) and the softmax denominator sums the fp16-ROUNDED probabilities in FP32, as the reference's tensor-core row
    sum does in every instantiation (qk_int_sv_f16_cuda_sm80.cu:313-320).  "fp16+fp32" differs from "fp32" only in the q scale groups
    the reference pairs with it (WARPQ = 16 at head_dim 128, core.py:602-604); "fp16" additionally honours ``smooth_v``."""
    if torch.compiler.is_compiling():
        return _compiled_call("fp16", q, k, v, tensor_layout, is_causal, qk_quant_gran, sm_scale, pv_accum_dtype, smooth_k, smooth_v, return_lse, kwargs)
    dtype = q.dtype
    _check_inputs(q, k, v)
    assert qk_quant_gran in ["per_warp", "per_thread", "per_block"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    if pv_accum_dtype not in ("fp32", "fp16", "fp16+fp32"):
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")
    torch.cuda.set_device(v.device)
    q, k, v, head_dim_og = _pad_head_dim(q, k, v)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "Last dim of qkv must be contiguous."
    if sm_scale is None:
        sm_scale = head_dim_og ** -0.5
    if pv_accum_dtype in ["fp32", "fp16+fp32"] and smooth_v:
        warnings.warn(f"pv_accum_dtype is {pv_accum_dtype}, smooth_v will be ignored.")   # core.py:608-610
        smooth_v = False
    warpq = 16 if (q.size(-1) == 128 and pv_accum_dtype == "fp16+fp32") else 32              # core.py:602-604
    fused = _fused_prepass_wanted(k, tensor_layout, kwargs.get("fused_prepass"))
    # default route: Q is quantised inside the attention kernel (same bits, no INT8 copy of Q in HBM, one launch less)
    fuse_q = qk_quant_gran == "per_thread" and pv_accum_dtype != "fp16+fp32" and kwargs.get("fuse_q_quant", _FUSE_Q16_DEFAULT)
    n_split = 0
    if fuse_q:
        B_, Hq_, Lq_, _, _, _, _ = _dims(q, tensor_layout)
        n_split = _split_kv_plan(B_, Hq_, Lq_, _dims(k, tensor_layout)[2], is_causal, kwargs.get("split_kv"))
    # fp16 inputs on that route: the kernel reads V's rows in place, no V image and no V half of the pre-pass (core.py:613's
    # `v.to(torch.float16)` is the identity for them); same bits as the image route
    v_rows = fuse_q and not n_split and not smooth_v and _v_rows_wanted(q, k, v, tensor_layout, is_causal, kwargs.get("v_in_place"))
    v_in_prepass = fused and not smooth_v and k.shape == v.shape and not v_rows          # the fp16 image comes out of the same launch as K
    lse_correction, _, k_int8, k_scale, v_image, _, _ = _prepass_kv(q, k, v, tensor_layout, qk_quant_gran, 64, smooth_k, False, return_lse,
                                                                    fused, v_fp8=False, v_fp16=v_in_prepass)
    vm = None
    if smooth_v:     # pv_accum_dtype == "fp16": sub_mean + fused v_mean epilogue (core.py:617-619)
        v_image, vm = sub_mean(v, tensor_layout)
        vm = vm.float()
    elif not v_in_prepass and not v_rows:
        v_image = prep_v_fp16(v, tensor_layout)
    if fuse_q:
        if n_split:
            o, lse = _attn_fused_q_split(_aligned(q, 8), k_int8, v_image, None, k_scale, tensor_layout, is_causal, _sm_log2(sm_scale),
                                         n_split, return_lse, v_mean=vm)
        else:
            o, lse = _attn_fused_q(_aligned(q, 8), k_int8, v if v_rows else v_image, None, k_scale, tensor_layout, is_causal, _sm_log2(sm_scale),
                                   return_lse, v_mean=vm, v_rows=v_rows)
        return _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale)
    q_int8, q_scale, gran, q_warp, sm_log2 = _quant_q(q, qk_quant_gran, tensor_layout, warpq, sm_scale)
    o, lse = _attn_dense(False, q_int8, k_int8, v_image, None, q_scale, k_scale, dtype, tensor_layout, is_causal,
                         gran, q_warp, sm_log2, pv_accum_dtype == "fp16+fp32", return_lse, v_mean=vm)
    return _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale)


def sageattn_qk_int8_pv_fp8_cuda(q, k, v, tensor_layout: str = "HND", is_causal: bool = False,
                                 qk_quant_gran: str = "per_thread", sm_scale: Optional[float] = None,
                                 pv_accum_dtype: str = "fp32+fp16", smooth_k: bool = True, smooth_v: bool = False,
                                 return_lse: bool = False, **kwargs: Any):
    """INT8 QK^T + FP8 (e4m3) PV (reference core.py:636-826).  "fp32+fp32" and "fp32+fp16" both
    run the two-level kernel with an FP32 tile buffer (gfx950's FP8 MFMA only writes FP32, so V
    keeps the full ``scale_max=448``; the reference's 2.25 is an FP16-accumulator artefact,
    core.py:805-807); "fp32" accumulates every tile straight into the output registers."""
    if torch.compiler.is_compiling():
        return _compiled_call("fp8", q, k, v, tensor_layout, is_causal, qk_quant_gran, sm_scale, pv_accum_dtype, smooth_k, smooth_v, return_lse, kwargs)
    dtype = q.dtype
    _check_inputs(q, k, v)
    assert qk_quant_gran in ["per_warp", "per_thread", "per_block"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    if pv_accum_dtype not in ("fp32", "fp32+fp32", "fp32+fp16"):
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")
    torch.cuda.set_device(v.device)
    q, k, v, head_dim_og = _pad_head_dim(q, k, v)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "Last dim of qkv must be contiguous."
    if sm_scale is None:
        sm_scale = head_dim_og ** -0.5
    if pv_accum_dtype in ("fp32+fp32", "fp32+fp16") and smooth_v:
        warnings.warn(f"pv_accum_dtype is '{pv_accum_dtype}', smooth_v will be ignored.")   # core.py:797-803
        smooth_v = False
    fuse_q = qk_quant_gran == "per_thread" and pv_accum_dtype != "fp32" and kwargs.get("fuse_q_quant", True)
    fused = _fused_prepass_wanted(k, tensor_layout, kwargs.get("fused_prepass")) and k.shape == v.shape
    folded = ops.fp8_folded(kwargs.get("fp8_scores"))
    if fuse_q:
        # default route: Q is quantised inside the attention kernel (same bits, no INT8 copy of Q in HBM).
        # (Running the V pre-pass on a side stream beside the K chain was measured and rejected: the two HBM-bound chains
        #  slow each other down and the cross-stream joins cost more than the launch gaps they hide, 956 -> 1130 us at C3.)
        lse_correction, _, k_int8, k_scale, v_image, v_scale, vm = _prepass_kv(q, k, v, tensor_layout, "per_thread", 64, smooth_k, smooth_v,
                                                                               return_lse, fused)
        B_, Hq_, Lq_, _, _, _, _ = _dims(q, tensor_layout)
        n_split = _split_kv_plan(B_, Hq_, Lq_, _dims(k, tensor_layout)[2], is_causal, kwargs.get("split_kv"), auto_default=False)
        if n_split:
            o, lse = _attn_fused_q_split(_aligned(q, 8), k_int8, v_image, v_scale, k_scale, tensor_layout, is_causal,
                                         _sm_log2(sm_scale), n_split, return_lse, v_mean=vm, folded_scores=folded)
        else:
            o, lse = _attn_fused_q(_aligned(q, 8), k_int8, v_image, v_scale, k_scale, tensor_layout, is_causal, _sm_log2(sm_scale),
                                   return_lse, v_mean=vm, folded_scores=folded)
        return _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale)
    lse_correction, _, k_int8, k_scale, v_image, v_scale, vm = _prepass_kv(q, k, v, tensor_layout, qk_quant_gran, 64, smooth_k, smooth_v,
                                                                           return_lse, fused)
    q_int8, q_scale, gran, q_warp, sm_log2 = _quant_q(q, qk_quant_gran, tensor_layout, 32, sm_scale)
    o, lse = _attn_dense(True, q_int8, k_int8, v_image, v_scale, q_scale, k_scale, dtype, tensor_layout, is_causal,
                         gran, q_warp, sm_log2, pv_accum_dtype != "fp32", return_lse, v_mean=vm, folded_scores=folded)
    return _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale)


def sageattn_qk_int8_pv_fp8_cuda_sm90(q, k, v, tensor_layout: str = "HND", is_causal: bool = False,
                                      qk_quant_gran: str = "per_thread", sm_scale: Optional[float] = None,
                                      pv_accum_dtype: str = "fp32+fp32", smooth_k: bool = True, return_lse: bool = False,
                                      **kwargs: Any):
    """Kept for drop-in (reference core.py:829-996).  Same gfx950 kernel and two-level accumulation as
    ``sageattn_qk_int8_pv_fp8_cuda``, with the sm90 kernels' scale groups: q per 16 rows (per-warp) or the 8 per-thread
    slots of every 16 rows, k per 128 keys (core.py:964-970: BLKQ=64, WARPQ=16, BLKK=128, WARPK=128)."""
    if pv_accum_dtype == "fp32":
        raise NotImplementedError("Please use pv_accum_dtype='fp32+fp32' for sm90.")   # core.py:985-986
    if pv_accum_dtype != "fp32+fp32":
        raise ValueError(f"Unsupported pv_accum_dtype: {pv_accum_dtype}")
    if torch.compiler.is_compiling():
        return _compiled_call("sm90", q, k, v, tensor_layout, is_causal, qk_quant_gran, sm_scale, pv_accum_dtype, smooth_k, False, return_lse, kwargs)
    dtype = q.dtype
    _check_inputs(q, k, v)
    assert qk_quant_gran in ["per_warp", "per_thread"], "qk_quant_gran must be either 'per_warp' or 'per_thread'."
    torch.cuda.set_device(v.device)
    q, k, v, head_dim_og = _pad_head_dim(q, k, v)
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1, "Last dim of qkv must be contiguous."
    if sm_scale is None:
        sm_scale = head_dim_og ** -0.5
    fused = _fused_prepass_wanted(k, tensor_layout, kwargs.get("fused_prepass")) and k.shape == v.shape
    lse_correction, _, k_int8, k_scale, v_image, v_scale, _ = _prepass_kv(q, k, v, tensor_layout, qk_quant_gran, 128, smooth_k, False,
                                                                          return_lse, fused)
    q_int8, q_scale, gran, q_warp, sm_log2 = _quant_q(q, qk_quant_gran, tensor_layout, 16, sm_scale, blkk=128)
    o, lse = _attn_dense(True, q_int8, k_int8, v_image, v_scale, q_scale, k_scale, dtype, tensor_layout, is_causal,
                         gran, q_warp, sm_log2, True, return_lse, folded_scores=ops.fp8_folded(kwargs.get("fp8_scores")))
    return _finish(o, lse, head_dim_og, return_lse, smooth_k, lse_correction, sm_scale)
