"""attn_qk_int8_per_block_causal.forward (sageattention/triton/attn_qk_int8_per_block_causal.py:124)."""
from sageattention_amd.kernel_api import forward_causal as forward      # noqa: F401
