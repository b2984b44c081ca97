"""attn_qk_int8_block_varlen.forward (sageattention/triton/attn_qk_int8_block_varlen.py:123)."""
import torch

from sageattention_amd import kernel_api as _k


def forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale, cu_seqlens_k_scale, output_dtype=torch.float16):
    return _k.forward_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, q_scale, k_scale, cu_seqlens_q_scale, cu_seqlens_k_scale,
                             output_dtype=output_dtype)
