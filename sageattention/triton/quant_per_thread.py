"""quant_per_thread.per_thread_int8 (sageattention/triton/quant_per_thread.py:154)."""
from sageattention_amd.quant import per_thread_int8      # noqa: F401
