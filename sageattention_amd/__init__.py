"""sageattention_amd -- SageAttention's quantized fused attention, hand-written for MI355X (gfx950).

Exports the reference's six public names (``sageattention/__init__.py:1-5``).
"""
from .core import sageattn, sageattn_varlen
from .core import sageattn_qk_int8_pv_fp16_triton
from .core import sageattn_qk_int8_pv_fp16_cuda
from .core import sageattn_qk_int8_pv_fp8_cuda
from .core import sageattn_qk_int8_pv_fp8_cuda_sm90

__all__ = [
    "sageattn", "sageattn_varlen", "sageattn_qk_int8_pv_fp16_triton", "sageattn_qk_int8_pv_fp16_cuda",
    "sageattn_qk_int8_pv_fp8_cuda", "sageattn_qk_int8_pv_fp8_cuda_sm90",
]
__version__ = "0.1.0"
