"""``sageattention.core`` of the reference (core.py): the public entry points, from the gfx950 implementation."""
from sageattention_amd.core import (sageattn, sageattn_varlen, sageattn_qk_int8_pv_fp16_triton, sageattn_qk_int8_pv_fp16_cuda,   # noqa: F401
                                    sageattn_qk_int8_pv_fp8_cuda, sageattn_qk_int8_pv_fp8_cuda_sm90)
