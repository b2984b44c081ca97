"""quant_per_block.per_block_int8 (sageattention/triton/quant_per_block.py:49): the Triton rounding convention."""
from typing import Optional

from sageattention_amd.quant import per_block_int8 as _per_block_int8


def per_block_int8(q, k, km=None, BLKQ: int = 128, BLKK: int = 64, sm_scale: Optional[float] = None, tensor_layout: str = "HND"):
    return _per_block_int8(q, k, km=km, BLKQ=BLKQ, BLKK=BLKK, sm_scale=sm_scale, tensor_layout=tensor_layout, quantization_backend="triton")
