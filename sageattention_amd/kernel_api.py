"""The reference's KERNEL-LEVEL Python entry points, on gfx950: the ``forward`` functions of its Triton attention modules
(sageattention/triton/attn_qk_int8_per_block.py:130, attn_qk_int8_per_block_causal.py:124, attn_qk_int8_block_varlen.py:123,
attn_qk_int8_per_block_causal_varlen.py:138) -- INT8 q / k with their per-block scales and the fp16 value tensor in, output (and LSE) out,
what the reference's own bench script times (bench/bench_qk_int8_pv_fp16_triton.py) -- with the reference's signatures.  ``sageattention/triton/*``
re-exports them under the reference's module names.  No Triton here: the same HIP kernel family runs them (C ABI
``sage_attn_qk_int8_pv_f16_vrows`` -- the value rows are read in place, as the reference's kernels read them -- ``_masked`` and ``_varlen``)."""
from typing import Optional

import torch

from . import _cabi, ops
from .quant import _cu_blocks, prep_v_fp16, prep_v_fp16_varlen

_LSE_UNUSED = None


def _scales(t: torch.Tensor, shape) -> torch.Tensor:
    """fp32 contiguous ``shape`` (the reference's quantisers return that; its bench script hands fp16 ``[..., 1]`` tensors, which Triton loads as they are)."""
    return t.reshape(shape).to(torch.float32).contiguous()


def _no_lse(device) -> torch.Tensor:
    return torch.empty([0], dtype=torch.float32, device="cpu")      # attn_qk_int8_per_block.py:164-167


def forward(q, k, v, q_scale, k_scale, tensor_layout: str = "HND", attn_mask: Optional[torch.Tensor] = None,
            output_dtype: torch.dtype = torch.float16, return_lse: bool = False, is_causal: bool = False):
    """attn_qk_int8_per_block.forward (``is_causal=True``: attn_qk_int8_per_block_causal.forward).  q, k INT8 with ``sm_scale * log2e`` folded into
    ``q_scale`` by the quantiser (quant_per_block.py:87); v fp16; returns ``(o, lse)`` with the LSE in log2 units as the reference's kernels
    leave it (``lse / 1.44269504`` is the caller's, core.py:328-329)."""
    assert q.is_cuda and q.dtype == torch.int8 and k.dtype == torch.int8, "q and k are the INT8 tensors of per_block_int8"
    assert v.dtype == torch.float16, "v must be fp16 (core.py:297-298 converts before the kernel)"
    assert output_dtype in (torch.float16, torch.bfloat16)
    assert tensor_layout in ("HND", "NHD")
    seq = 2 if tensor_layout == "HND" else 1
    B, Hq, Hkv, Lq, Lk = q.shape[0], q.shape[3 - seq], k.shape[3 - seq], q.shape[seq], k.shape[seq]
    assert Hq % Hkv == 0, "num_qo_heads must be divisible by num_kv_heads"
    assert q.shape[-1] in (64, 128), "head_dim 64 or 128 (core.py:260-271 pads before quantising)"
    q, k = q.contiguous(), k.contiguous()
    qs = _scales(q_scale, (B, Hq, (Lq + 127) // 128))
    ks = _scales(k_scale, (B, Hkv, (Lk + 63) // 64))
    if attn_mask is not None:
        assert not is_causal, "Mask should be None for causal attention."
        from .core import _attn_masked
        o, lse = _attn_masked(q, k, prep_v_fp16(v, tensor_layout), qs, ks, attn_mask, output_dtype, tensor_layout, return_lse)
        return o, (lse if return_lse else _no_lse(q.device))
    o = torch.empty(q.shape, dtype=output_dtype, device=q.device)
    in_place = v.stride(-1) == 1 and v.data_ptr() % 16 == 0 and all(s % 8 == 0 for s in v.stride()[:-1])
    lse = ops.qk_int8_sv_f16_attn_impl(q, k, v if in_place else prep_v_fp16(v, tensor_layout), o, qs, ks, None, 0 if tensor_layout == "NHD" else 1,
                                       int(is_causal), _cabi.GRAN_PER_BLOCK, 128, 1.0, _cabi.PV_ACCUM_TRITON, int(return_lse))
    return o, (lse if return_lse else _no_lse(q.device))


def forward_causal(q, k, v, q_scale, k_scale, tensor_layout: str = "HND", output_dtype: torch.dtype = torch.float16, return_lse: bool = False):
    """attn_qk_int8_per_block_causal.forward."""
    return forward(q, k, v, q_scale, k_scale, tensor_layout=tensor_layout, output_dtype=output_dtype, return_lse=return_lse, is_causal=True)


def forward_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale, cu_seqlens_k_scale,
                   output_dtype: torch.dtype = torch.float16, is_causal: bool = False):
    """attn_qk_int8_block_varlen.forward (``is_causal=True``: attn_qk_int8_per_block_causal_varlen.forward): packed ``[sum L, H, D]`` INT8 q / k
    with the block-major scale tensors and prefix arrays of quant_per_block_varlen.per_block_int8, fp16 v; returns ``o``."""
    assert q.is_cuda and q.dtype == torch.int8 and k.dtype == torch.int8 and v.dtype == torch.float16
    assert output_dtype in (torch.float16, torch.bfloat16)
    Hq, Hkv, D = q.shape[1], k.shape[1], q.shape[2]
    assert Hq % Hkv == 0 and D in (64, 128)
    q, k = q.contiguous(), k.contiguous()
    cu_q, cu_k = cu_seqlens_q.to(torch.int32).contiguous(), cu_seqlens_k.to(torch.int32).contiguous()
    cu_qs, cu_ks = cu_seqlens_q_scale.to(torch.int32).contiguous(), cu_seqlens_k_scale.to(torch.int32).contiguous()
    nseq = cu_q.shape[0] - 1
    qs = q_scale.reshape(-1, Hq).to(torch.float32).contiguous()
    ks = k_scale.reshape(-1, Hkv).to(torch.float32).contiguous()
    # the V tile image of the packed rows (64-key tiles per sequence, at the k-scale prefix array); every sequence is at most all packed tokens long
    v_image = prep_v_fp16_varlen(v, cu_k, cu_ks, k.shape[0], ntiles=(k.shape[0] + 63) // 64 + nseq)
    order = torch.argsort(cu_q[1:] - cu_q[:-1], descending=True).to(torch.int32)
    o = torch.empty(q.shape, dtype=output_dtype, device=q.device)
    code = _cabi.DTYPE_F16 if output_dtype == torch.float16 else _cabi.DTYPE_BF16
    p = lambda t: None if t is None else t.data_ptr()
    rc = _cabi.load().sage_attn_qk_int8_pv_f16_varlen(
        p(q), p(k), p(v_image), p(o), p(qs), p(ks), p(cu_q), p(cu_k), p(cu_qs), p(cu_ks), p(order), None, None, 0,
        nseq, int(max_seqlen_q), Hq, Hkv, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1), o.stride(0), o.stride(1),
        int(is_causal), 1.0, _cabi.PV_ACCUM_TRITON, code, torch._C._cuda_getCurrentRawStream(o.device.index), None)
    _cabi.check(rc, "sage_attn_qk_int8_pv_f16_varlen")
    return o


def forward_varlen_causal(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale, cu_seqlens_k_scale,
                          output_dtype: torch.dtype = torch.float16):
    """attn_qk_int8_per_block_causal_varlen.forward."""
    return forward_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale, cu_seqlens_k_scale,
                          output_dtype=output_dtype, is_causal=True)


def per_block_int8_varlen_ref(q, k, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, BLKQ: int = 128, BLKK: int = 64,
                              sm_scale: Optional[float] = None):
    """quant_per_block_varlen.per_block_int8 with the reference's signature and return values
    ``(q_int8, q_scale, k_int8, k_scale, cu_seqlens_q_scale, cu_seqlens_k_scale)`` (quant_per_block_varlen.py:60-104; k comes smoothed)."""
    from .quant import per_block_int8_varlen
    return per_block_int8_varlen(q, k, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, BLKQ=BLKQ, BLKK=BLKK, sm_scale=sm_scale)
