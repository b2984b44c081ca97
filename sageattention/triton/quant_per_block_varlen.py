"""quant_per_block_varlen.per_block_int8 (sageattention/triton/quant_per_block_varlen.py:60)."""
from sageattention_amd.kernel_api import per_block_int8_varlen_ref as per_block_int8      # noqa: F401
