"""attn_qk_int8_per_block.forward (sageattention/triton/attn_qk_int8_per_block.py:130)."""
import torch

from sageattention_amd import kernel_api as _k


def forward(q, k, v, q_scale, k_scale, tensor_layout="HND", attn_mask=None, output_dtype=torch.float16, return_lse=False):
    return _k.forward(q, k, v, q_scale, k_scale, tensor_layout=tensor_layout, attn_mask=attn_mask, output_dtype=output_dtype, return_lse=return_lse)
