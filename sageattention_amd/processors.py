"""Callers of the hot path: an SDPA-signature adapter and a diffusers-style attention processor.

The reference plugs ``sageattn`` into models in two ways (README "plug-and-play"; ``example/modify_model/``):
replacing ``F.scaled_dot_product_attention`` and installing a per-model attention processor
(``modify_wan.py:8-99``, ``modify_mochi.py:7-110``, ``modify_ltx.py:11-86``) whose ``__call__`` projects q/k/v,
normalises, applies rotary embeddings, calls ``attn_func(q, k, v, attn_mask=, dropout_p=0.0, is_causal=False)`` on
``[B, H, L, D]`` tensors and projects out.  ``diffusers`` is third-party and absent from this build, so the
processor here is duck-typed on the attributes those processors use (``to_q/to_k/to_v``, ``norm_q/norm_k``,
``heads``, ``to_out``); it keeps activations in ``[B, L, H, D]`` (``tensor_layout="NHD"``) so neither the
transposes of the reference processors nor their copies are needed: the kernels take strides.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .core import sageattn, sageattn_qk_int8_pv_fp16_triton


def sdpa(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_mask: Optional[torch.Tensor] = None,
         dropout_p: float = 0.0, is_causal: bool = False, scale: Optional[float] = None, enable_gqa: bool = False,
         tensor_layout: str = "HND") -> torch.Tensor:
    """``torch.nn.functional.scaled_dot_product_attention`` signature on the gfx950 kernels.

    Without a mask: ``sageattn`` (INT8 QK^T + FP8 PV, two-level accumulation).  With ``attn_mask`` (bool or additive,
    broadcastable to ``[B, H, Lq, Lk]``): the Triton-named API, the only reference entry point that honours a mask
    (core.py:313-324) -- the reference's ``sageattn`` silently drops ``attn_mask``; this adapter does not.
    Dropout is not part of the reference's path and is rejected."""
    if dropout_p != 0.0:
        raise NotImplementedError("sageattention has no dropout (the reference's callers pass dropout_p=0.0)")
    if attn_mask is not None:
        if is_causal:
            raise ValueError("pass either attn_mask or is_causal, as with scaled_dot_product_attention")
        return sageattn_qk_int8_pv_fp16_triton(query, key, value, tensor_layout=tensor_layout, is_causal=False,
                                               attn_mask=attn_mask, sm_scale=scale)
    return sageattn(query, key, value, tensor_layout=tensor_layout, is_causal=is_causal, sm_scale=scale)


class SageAttnProcessor:
    """Generic diffusers-style attention processor: self- and cross-attention (Lq != Lk), optional q/k norms,
    optional rotary embedding callback, ``[B, L, H, D]`` activations straight into the NHD kernels."""

    def __init__(self, attn_func: Optional[Callable] = None, apply_rotary: Optional[Callable] = None):
        self.attn_func = attn_func or sdpa
        self.apply_rotary = apply_rotary

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, rotary_emb=None, **kwargs) -> torch.Tensor:
        context = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        query, key, value = attn.to_q(hidden_states), attn.to_k(context), attn.to_v(context)
        if getattr(attn, "norm_q", None) is not None:
            query = attn.norm_q(query)
        if getattr(attn, "norm_k", None) is not None:
            key = attn.norm_k(key)
        heads = attn.heads
        query = query.unflatten(2, (heads, -1))             # [B, L, H, D]: the NHD layout, no transpose
        key = key.unflatten(2, (heads, -1))
        value = value.unflatten(2, (heads, -1))
        if rotary_emb is not None and encoder_hidden_states is None:
            if self.apply_rotary is None:
                raise ValueError("rotary_emb given but the processor has no apply_rotary callback")
            query, key = self.apply_rotary(query, rotary_emb), self.apply_rotary(key, rotary_emb)
        out = self.attn_func(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False,
                             tensor_layout="NHD")
        out = out.flatten(2, 3).type_as(hidden_states)
        out = attn.to_out[0](out)
        if len(attn.to_out) > 1:
            out = attn.to_out[1](out)
        return out


def set_sage_attention(model: torch.nn.Module, attn_func: Optional[Callable] = None,
                       apply_rotary: Optional[Callable] = None, predicate: Optional[Callable] = None) -> int:
    """Install :class:`SageAttnProcessor` on every sub-module that has a ``processor`` attribute and the projection
    layers the processor needs (what ``set_sage_attn_wan/_mochi/_ltx`` do for one model family each,
    modify_wan.py:102-109).  Returns the number of modules patched."""
    n = 0
    for name, mod in model.named_modules():
        if all(hasattr(mod, a) for a in ("processor", "to_q", "to_k", "to_v", "to_out", "heads")):
            if predicate is None or predicate(name, mod):
                mod.processor = SageAttnProcessor(attn_func, apply_rotary)
                n += 1
    return n
