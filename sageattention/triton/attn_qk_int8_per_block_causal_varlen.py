"""attn_qk_int8_per_block_causal_varlen.forward (sageattention/triton/attn_qk_int8_per_block_causal_varlen.py:138)."""
from sageattention_amd.kernel_api import forward_varlen_causal as forward      # noqa: F401
