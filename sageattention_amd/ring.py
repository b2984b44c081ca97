"""Sequence-parallel (ring) attention built on ``sageattn(..., return_lse=True)``.

The reference exposes ``return_lse`` (core.py:782-786, 823-826) as its long-context hook; its
``example/parallel_sageattn_cogvideo.py`` hands ``sageattn`` to xfuser's ring attention, which runs the
kernel against one K/V shard at a time and combines the partial results by their log-sum-exp.  This module
is that caller written for MI355X: one process per GPU, K/V shards travel around the ring as point-to-point
``isend/irecv`` (RCCL over xGMI, posted before the local attention so the copy overlaps the kernel), and the
partial states are merged by the HIP kernel behind ``sage_merge_states`` (FP32 running state, one rounding).

Sharding: rank ``r`` of ``W`` holds the contiguous token range ``[r L/W, (r+1) L/W)`` of q, k and v.
Causal masking at shard granularity: a shard from an earlier rank is attended in full, the rank's own shard
causally, shards from later ranks are skipped -- so rank 0 does one step of work and rank W-1 does W.
``shard_order="zigzag"`` balances that: the sequence is cut into 2W chunks and rank ``r`` holds chunks ``r`` and
``2W-1-r`` (:func:`zigzag_shard`); every rank then does the same 2c^2 of causal work per step.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import _cabi
from .quant import _dims, _p, _stream


@torch.compiler.disable
def merge_states(o_acc: torch.Tensor, lse_acc: torch.Tensor, o_new: torch.Tensor, lse_new: torch.Tensor,
                 tensor_layout: str = "HND", first: bool = False, out: Optional[torch.Tensor] = None) -> None:
    """In place: fold the partial state ``(o_new [fp16/bf16], lse_new [B,H,L] fp32, natural log)`` into the running
    FP32 state ``(o_acc [B,H,L,D], lse_acc [B,H,L])``.  ``first`` initialises the state; ``out`` (same shape, layout
    and dtype as ``o_new``) additionally receives the merged output -- pass it on the last step."""
    B, H, L, D, n_sb, n_sh, n_sl = _dims(o_new, tensor_layout)
    assert o_acc.shape == (B, H, L, D) and o_acc.dtype == torch.float32 and o_acc.is_contiguous()
    assert lse_acc.shape == (B, H, L) and lse_acc.dtype == torch.float32 and lse_acc.is_contiguous()
    assert lse_new.shape == (B, H, L) and lse_new.dtype == torch.float32
    lse_new = lse_new.contiguous()
    assert o_new.dtype in (torch.float16, torch.bfloat16) and o_new.stride(-1) == 1
    o_sb = o_sh = o_sl = 0
    if out is not None:
        assert out.shape == o_new.shape and out.dtype == o_new.dtype and out.stride(-1) == 1
        _, _, _, _, o_sb, o_sh, o_sl = _dims(out, tensor_layout)
    code = _cabi.DTYPE_F16 if o_new.dtype == torch.float16 else _cabi.DTYPE_BF16
    rc = _cabi.load().sage_merge_states(_p(o_acc), _p(lse_acc), _p(o_new), _p(lse_new), _p(out), B, H, L, D,
                                        n_sb, n_sh, n_sl, o_sb, o_sh, o_sl, code, int(first), _stream(o_new))
    _cabi.check(rc, "sage_merge_states")


def shard_schedule(rank: int, world: int, is_causal: bool):
    """[(step, kv_chunk, mode)] for one rank: at step s the rank holds the K/V shard of rank (rank - s) mod world;
    mode is "causal" (own shard, causal attention), "full", or "skip" (a later shard under causal masking)."""
    sched = []
    for s in range(world):
        j = (rank - s) % world
        if not is_causal:
            mode = "full"
        elif j == rank:
            mode = "causal"
        else:
            mode = "full" if j < rank else "skip"
        sched.append((s, j, mode))
    return sched


def zigzag_shard(x: torch.Tensor, rank: int, world: int, tensor_layout: str = "HND") -> torch.Tensor:
    """The zig-zag shard of a full-sequence tensor: chunks ``rank`` and ``2 world - 1 - rank`` of ``2 world``."""
    dim = 2 if tensor_layout == "HND" else 1
    L = x.shape[dim]
    assert L % (2 * world) == 0, "zig-zag sharding needs the sequence length to be a multiple of 2 * world"
    c = L // (2 * world)
    lo, hi = x.narrow(dim, rank * c, c), x.narrow(dim, (2 * world - 1 - rank) * c, c)
    return torch.cat([lo, hi], dim=dim)


def zigzag_schedule(rank: int, world: int):
    """Causal work of one rank under zig-zag sharding: [(step, kv_rank, [(q_part, kv_part, mode), ...])] with parts
    "lo" / "hi" / "all" (the rank's first / second chunk / both) -- two or three kernel calls at step 0, one per step
    afterwards, 2 c^2 score elements per step on every rank."""
    sched = []
    for s in range(world):
        j = (rank - s) % world
        if j == rank:
            calls = [("lo", "lo", "causal"), ("hi", "lo", "full"), ("hi", "hi", "causal")]
        elif j < rank:
            calls = [("all", "lo", "full")]          # both of our chunks come after chunk j, before chunk 2W-1-j
        else:
            calls = [("hi", "all", "full")]          # only our late chunk sees rank j's chunks, and sees both in full
        sched.append((s, j, calls))
    return sched


def _ring_zigzag_causal(q, k, v, group, tensor_layout, sm_scale, return_lse, attn_fn, merge_fn, world, rank, kwargs):
    import torch.distributed as dist
    dim = 2 if tensor_layout == "HND" else 1
    B, H, L2, D, _, _, _ = _dims(q, tensor_layout)
    assert L2 % 2 == 0
    c = L2 // 2
    part = lambda t, name: t if name == "all" else t.narrow(dim, 0 if name == "lo" else c, c)
    acc = {n: torch.empty((B, H, c, D), dtype=torch.float32, device=q.device) for n in ("lo", "hi")}
    lse = {n: torch.empty((B, H, c), dtype=torch.float32, device=q.device) for n in ("lo", "hi")}
    out = torch.empty_like(q)
    sched = zigzag_schedule(rank, world)
    # per q half: the ordered list of (step, call index) that touch it, to know the first and the last merge
    touches = {"lo": [], "hi": []}
    for s, _, calls in sched:
        for ci, (qp, _, _) in enumerate(calls):
            for n in (("lo", "hi") if qp == "all" else (qp,)):
                touches[n].append((s, ci))
    k_cur, v_cur = k.contiguous(), v.contiguous()
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    if group is not None and world > 1:
        nxt, prv = dist.get_global_rank(group, nxt), dist.get_global_rank(group, prv)
    for s, j, calls in sched:
        reqs, k_nxt, v_nxt = [], None, None
        if s + 1 < world:
            k_nxt, v_nxt = torch.empty_like(k_cur), torch.empty_like(v_cur)
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, k_cur, nxt, group), dist.P2POp(dist.isend, v_cur, nxt, group),
                                           dist.P2POp(dist.irecv, k_nxt, prv, group), dist.P2POp(dist.irecv, v_nxt, prv, group)])
        for ci, (qp, kp, mode) in enumerate(calls):
            o_s, lse_s = attn_fn(part(q, qp), part(k_cur, kp), part(v_cur, kp), tensor_layout=tensor_layout,
                                 is_causal=(mode == "causal"), sm_scale=sm_scale, return_lse=True, **kwargs)
            for hi_idx, n in enumerate(("lo", "hi")):
                if qp not in ("all", n):
                    continue
                o_n = o_s if qp != "all" else o_s.narrow(dim, hi_idx * c, c)
                l_n = lse_s if qp != "all" else lse_s.narrow(2, hi_idx * c, c)
                merge_fn(acc[n], lse[n], o_n, l_n, tensor_layout=tensor_layout, first=((s, ci) == touches[n][0]),
                         out=out.narrow(dim, hi_idx * c, c) if (s, ci) == touches[n][-1] else None)
        for r in reqs:
            r.wait()
        if k_nxt is not None:
            k_cur, v_cur = k_nxt, v_nxt
    return (out, torch.cat([lse["lo"], lse["hi"]], dim=2)) if return_lse else out


def ring_sageattn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, group=None, tensor_layout: str = "HND",
                  is_causal: bool = False, sm_scale: Optional[float] = None, return_lse: bool = False,
                  attn_fn: Optional[Callable] = None, merge_fn: Optional[Callable] = None,
                  shard_order: str = "contiguous", **kwargs):
    """Attention over a sequence sharded across the ranks of ``group`` (see the module docstring).

    ``attn_fn(q, k, v, tensor_layout=, is_causal=, sm_scale=, return_lse=True, **kwargs) -> (o, lse)`` defaults to
    ``sageattn``; ``merge_fn`` defaults to :func:`merge_states` (both are seams for host-logic tests).
    ``shard_order``: "contiguous", or "zigzag" (inputs sharded with :func:`zigzag_shard`; only changes the causal case,
    without a mask the two orders do the same work)."""
    import torch.distributed as dist
    from .core import sageattn
    attn_fn = attn_fn or sageattn
    merge_fn = merge_fn or merge_states
    assert shard_order in ("contiguous", "zigzag")
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    B, H, L, D, _, _, _ = _dims(q, tensor_layout)
    if sm_scale is None:
        sm_scale = D ** -0.5
    if shard_order == "zigzag" and is_causal:
        return _ring_zigzag_causal(q, k, v, group, tensor_layout, sm_scale, return_lse, attn_fn, merge_fn, world, rank, kwargs)

    sched = shard_schedule(rank, world, is_causal)
    active = [s for s, _, mode in sched if mode != "skip"]
    o_acc = torch.empty((B, H, L, D), dtype=torch.float32, device=q.device)
    lse_acc = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
    out = torch.empty_like(q)
    k_cur, v_cur = k.contiguous(), v.contiguous()
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    if group is not None and world > 1:
        nxt, prv = dist.get_global_rank(group, nxt), dist.get_global_rank(group, prv)

    for s, j, mode in sched:
        reqs, k_nxt, v_nxt = [], None, None
        if s + 1 < world:               # pass the shard on while this step computes
            k_nxt, v_nxt = torch.empty_like(k_cur), torch.empty_like(v_cur)
            ops = [dist.P2POp(dist.isend, k_cur, nxt, group), dist.P2POp(dist.isend, v_cur, nxt, group),
                   dist.P2POp(dist.irecv, k_nxt, prv, group), dist.P2POp(dist.irecv, v_nxt, prv, group)]
            reqs = dist.batch_isend_irecv(ops)
        if mode != "skip":
            o_s, lse_s = attn_fn(q, k_cur, v_cur, tensor_layout=tensor_layout, is_causal=(mode == "causal"),
                                 sm_scale=sm_scale, return_lse=True, **kwargs)
            merge_fn(o_acc, lse_acc, o_s, lse_s, tensor_layout=tensor_layout, first=(s == active[0]),
                     out=out if s == active[-1] else None)
        for r in reqs:
            r.wait()
        if k_nxt is not None:
            k_cur, v_cur = k_nxt, v_nxt
    return (out, lse_acc) if return_lse else out
