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


# ------------------------------------------------------------------------------------------------------------------
# Model-family processors (example/modify_model/modify_wan.py, modify_mochi.py, modify_ltx.py in the reference).
# Duck-typed on the attributes those families' attention modules expose (diffusers is not part of this build); all of
# them keep activations in [B, L, H, D] and hand the kernels strides (tensor_layout="NHD") instead of transposing.

def rotary_interleaved(x: torch.Tensor, freqs_cos: torch.Tensor, freqs_sin: torch.Tensor) -> torch.Tensor:
    """Rotary embedding on interleaved (even, odd) channel pairs of ``x`` [B, L, H, D]; ``freqs_*`` broadcast to x with the
    pair value repeated on both channels (Wan's convention, modify_wan.py:40-53: cos taken from even, sin from odd slots)."""
    x1, x2 = x.unflatten(-1, (-1, 2)).unbind(-1)
    cos, sin = freqs_cos[..., 0::2], freqs_sin[..., 1::2]
    return torch.stack((x1 * cos - x2 * sin, x1 * sin + x2 * cos), dim=-1).flatten(-2).type_as(x)


def rotary_halfwidth(x: torch.Tensor, freqs_cos: torch.Tensor, freqs_sin: torch.Tensor) -> torch.Tensor:
    """Mochi's convention (modify_mochi.py:50-58): ``freqs_*`` have D/2 channels, the rotation is computed in fp32."""
    xe, xo = x[..., 0::2].float(), x[..., 1::2].float()
    return torch.stack(((xe * freqs_cos - xo * freqs_sin).to(x.dtype), (xe * freqs_sin + xo * freqs_cos).to(x.dtype)), dim=-1).flatten(-2)


def _qkv(attn, hidden_states, context):
    """q, k, v projections, also for modules whose projections were fused (``to_qkv`` / ``to_kv``)."""
    if getattr(attn, "fused_projections", False):
        if context is hidden_states and hasattr(attn, "to_qkv"):
            return attn.to_qkv(hidden_states).chunk(3, dim=-1)
        k, v = attn.to_kv(context).chunk(2, dim=-1)
        return attn.to_q(hidden_states), k, v
    return attn.to_q(hidden_states), attn.to_k(context), attn.to_v(context)


def _project_out(attn, x):
    x = attn.to_out[0](x)
    return attn.to_out[1](x) if len(attn.to_out) > 1 else x


class SageWanAttnProcessor:
    """Wan 2.x (T2V / I2V) attention: RMS-normalised q/k over the full width, interleaved rotary embedding on self-attention,
    and -- image-to-video -- the encoder context split into image tokens (all but the last ``text_context_length``) that
    get their own k/v projections and a SECOND attention call whose output is added (modify_wan.py:27-29,72-90)."""

    def __init__(self, attn_func: Optional[Callable] = None, text_context_length: int = 512):
        self.attn_func = attn_func or sdpa
        self.text_context_length = text_context_length

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, rotary_emb=None, **kwargs) -> torch.Tensor:
        image_context = None
        if getattr(attn, "add_k_proj", None) is not None and encoder_hidden_states is not None:
            n_img = encoder_hidden_states.shape[1] - self.text_context_length
            image_context, encoder_hidden_states = encoder_hidden_states[:, :n_img], encoder_hidden_states[:, n_img:]
        context = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        query, key, value = _qkv(attn, hidden_states, context)
        query, key = attn.norm_q(query), attn.norm_k(key)
        heads = attn.heads
        query, key, value = (t.unflatten(2, (heads, -1)) for t in (query, key, value))
        if rotary_emb is not None:
            query, key = rotary_interleaved(query, *rotary_emb), rotary_interleaved(key, *rotary_emb)
        out = self.attn_func(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False, tensor_layout="NHD")
        out = out.flatten(2, 3).type_as(query)
        if image_context is not None:
            key_img = attn.norm_added_k(attn.add_k_proj(image_context)).unflatten(2, (heads, -1))
            value_img = attn.add_v_proj(image_context).unflatten(2, (heads, -1))
            out_img = self.attn_func(query, key_img, value_img, attn_mask=None, dropout_p=0.0, is_causal=False, tensor_layout="NHD")
            out = out + out_img.flatten(2, 3).type_as(query)
        return _project_out(attn, out)


class SageMochiAttnProcessor:
    """Mochi joint attention (modify_mochi.py:15-110): video tokens and the VALID prompt tokens of each sample attend jointly.
    The reference loops over the batch, gathers the valid prompt tokens, calls the attention once per sample and pads the
    result back.  With the default ``attn_func=None`` this processor packs all samples into ONE variable-length call
    (``sageattn_varlen``: per-sample lengths Lv + n_valid_i) -- the ragged batch is what that kernel exists for; a custom
    ``attn_func`` is called once per sample like the reference does.  Returns ``(hidden_states, encoder_hidden_states)``."""

    def __init__(self, attn_func: Optional[Callable] = None):
        self.attn_func = attn_func

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, attention_mask: torch.Tensor,
                 image_rotary_emb=None, **kwargs):
        heads = attn.heads
        query, key, value = (p(hidden_states).unflatten(2, (heads, -1)) for p in (attn.to_q, attn.to_k, attn.to_v))
        if getattr(attn, "norm_q", None) is not None:
            query = attn.norm_q(query)
        if getattr(attn, "norm_k", None) is not None:
            key = attn.norm_k(key)
        eq, ek, ev = (p(encoder_hidden_states).unflatten(2, (heads, -1)) for p in (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj))
        if getattr(attn, "norm_added_q", None) is not None:
            eq = attn.norm_added_q(eq)
        if getattr(attn, "norm_added_k", None) is not None:
            ek = attn.norm_added_k(ek)
        if image_rotary_emb is not None:
            query, key = rotary_halfwidth(query, *image_rotary_emb), rotary_halfwidth(key, *image_rotary_emb)
        B, Lv, Le = query.shape[0], query.shape[1], eq.shape[1]
        valid = attention_mask.reshape(B, Le) != 0
        out = query.new_zeros(B, Lv + Le, heads, query.shape[-1])
        if self.attn_func is None:
            from .core import sageattn_varlen
            n_valid = valid.sum(dim=1)
            lens = (n_valid + Lv).to(torch.int32)
            cu = torch.zeros(B + 1, dtype=torch.int32, device=query.device)
            cu[1:] = lens.cumsum(0)
            idx = [torch.nonzero(valid[b], as_tuple=False).flatten() for b in range(B)]
            pack = lambda t, e: torch.cat([torch.cat((t[b], e[b, idx[b]]), dim=0) for b in range(B)], dim=0)
            max_len = int(lens.max().item())
            o = sageattn_varlen(pack(query, eq), pack(key, ek), pack(value, ev), cu, cu, max_len, max_len, is_causal=False)
            starts = cu.tolist()
            for b in range(B):
                out[b, :starts[b + 1] - starts[b]] = o[starts[b]:starts[b + 1]]
        else:
            for b in range(B):
                ib = torch.nonzero(valid[b], as_tuple=False).flatten()
                qb, kb, vb = (torch.cat((t[b:b + 1], e[b:b + 1, ib]), dim=1) for t, e in ((query, eq), (key, ek), (value, ev)))
                ob = self.attn_func(qb, kb, vb, dropout_p=0.0, is_causal=False, tensor_layout="NHD")
                out[b, :ob.shape[1]] = ob[0]
        out = out.flatten(2, 3)
        hidden, enc = out.split_with_sizes((Lv, Le), dim=1)
        hidden = _project_out(attn, hidden)
        if hasattr(attn, "to_add_out"):
            enc = attn.to_add_out(enc)
        return hidden, enc


class SageLTXAttnProcessor:
    """LTX-Video attention (modify_ltx.py:27-81): q/k norms and the rotary embedding act on the full [B, L, H*D] width before
    the head split; the additive / boolean mask prepared by the module is honoured (the mask-capable kernel)."""

    def __init__(self, attn_func: Optional[Callable] = None, apply_rotary: Optional[Callable] = None):
        self.attn_func = attn_func or sdpa
        self.apply_rotary = apply_rotary

    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, image_rotary_emb=None, **kwargs) -> torch.Tensor:
        context = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        batch, ctx_len = context.shape[0], context.shape[1]
        if attention_mask is not None and hasattr(attn, "prepare_attention_mask"):
            attention_mask = attn.prepare_attention_mask(attention_mask, ctx_len, batch)
            attention_mask = attention_mask.view(batch, attn.heads, -1, attention_mask.shape[-1])
        query, key, value = attn.to_q(hidden_states), attn.to_k(context), attn.to_v(context)
        query, key = attn.norm_q(query), attn.norm_k(key)
        if image_rotary_emb is not None:
            rot = self.apply_rotary or getattr(attn, "apply_rotary_emb", None)
            if rot is None:
                raise ValueError("image_rotary_emb given but neither the processor nor the module has a rotary callback")
            query, key = rot(query, image_rotary_emb), rot(key, image_rotary_emb)
        heads = attn.heads
        query, key, value = (t.unflatten(2, (heads, -1)) for t in (query, key, value))
        out = self.attn_func(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False, tensor_layout="NHD")
        return _project_out(attn, out.flatten(2, 3).to(query.dtype))


def _blocks(model, names):
    for nm in names:
        if hasattr(model, nm):
            return list(getattr(model, nm))
    raise AttributeError(f"model has none of {names}")


def set_sage_attn_wan(model: torch.nn.Module, attn_func: Optional[Callable] = None) -> int:
    """``block.attn1`` (self-attention) of every Wan block, as modify_wan.py:102-109; ``attn2`` (text / image cross-attention)
    too when ``cross=True`` semantics are wanted, install :class:`SageWanAttnProcessor` on it yourself."""
    blocks = _blocks(model, ("blocks",))
    for blk in blocks:
        blk.attn1.processor = SageWanAttnProcessor(attn_func)
    return len(blocks)


def set_sage_attn_mochi(model: torch.nn.Module, attn_func: Optional[Callable] = None) -> int:
    """Every Mochi transformer block but the last (modify_mochi.py:116-119)."""
    blocks = _blocks(model, ("transformer_blocks",))[:-1]
    for blk in blocks:
        blk.attn1.processor = SageMochiAttnProcessor(attn_func)
    return len(blocks)


def set_sage_attn_ltx(model: torch.nn.Module, attn_func: Optional[Callable] = None, apply_rotary: Optional[Callable] = None) -> int:
    """``attn1`` of every LTX transformer block; the replaced processor is kept as ``origin_processor`` (modify_ltx.py:86-98)."""
    blocks = _blocks(model, ("transformer_blocks",))
    for blk in blocks:
        a = blk.attn1
        if not hasattr(a, "origin_processor"):
            a.origin_processor = a.get_processor() if hasattr(a, "get_processor") else getattr(a, "processor", None)
        proc = SageLTXAttnProcessor(attn_func, apply_rotary)
        if hasattr(a, "set_processor"):
            a.set_processor(proc)
        else:
            a.processor = proc
    return len(blocks)
