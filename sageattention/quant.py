"""``sageattention.quant`` of the reference (quant.py: the CUDA-backend quantisers), from the gfx950 implementation.  ``per_block_int8`` here is
the CUDA rounding convention (quant.py:22-103); the Triton one lives in ``sageattention.triton.quant_per_block`` as in the reference.
``per_channel_fp8`` / ``sub_mean`` return the gfx950 tile image of V in place of the reference's transposed-permuted tensor: both layouts are
private to (quantiser, kernel) pairs of one implementation (include/sage_gfx950.h)."""
from typing import Optional

from sageattention_amd.quant import per_warp_int8, sub_mean          # noqa: F401
from sageattention_amd.quant import per_channel_fp8 as _per_channel_fp8
from sageattention_amd.quant import per_block_int8 as _per_block_int8


def per_block_int8(q, k, km=None, BLKQ: int = 128, BLKK: int = 64, sm_scale: Optional[float] = None, tensor_layout: str = "HND"):
    return _per_block_int8(q, k, km=km, BLKQ=BLKQ, BLKK=BLKK, sm_scale=sm_scale, tensor_layout=tensor_layout, quantization_backend="cuda")


def per_channel_fp8(v, tensor_layout: str = "HND", scale_max: float = 448.0, smooth_v: bool = True):      # (the reference's defaults, quant.py:224-229)
    return _per_channel_fp8(v, tensor_layout=tensor_layout, scale_max=scale_max, smooth_v=smooth_v)
