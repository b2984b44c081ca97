"""Batch*kv-head sharding for multi-GPU runs.

The hot path is embarrassingly parallel over (batch, kv-head group, 128-row q-block): K, V, km
and v_scale of one (batch, kv-head) are needed only by that kv-head's q-heads.  So the unit of
sharding is a (batch, kv-head) pair together with its Hq/Hkv query heads; units are split
contiguously across ranks and there is NO collective on the data path -- one process per GPU,
each runs the unchanged single-GPU kernels on its slice (the reference has no parallelism of
its own either; its multi-GPU example delegates to xfuser, example/parallel_sageattn_cogvideo.py).
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_units`` for ``rank``; sizes differ by at most one."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_bh(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rank: int, world: int):
    """Slice HND tensors ``[B,H,L,D]`` to this rank's (batch, kv-head) units.

    Returns tensors reshaped to ``[1, units*g, L, D]`` / ``[1, units, L, D]`` (batch folded into
    heads, which the kernels treat identically) and the (lo, hi) unit range."""
    B, Hq, Lq, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    lo, hi = shard_range(B * Hkv, rank, world)
    qf = q.reshape(B * Hkv, g, Lq, D)[lo:hi].reshape(1, (hi - lo) * g, Lq, D)
    kf = k.reshape(B * Hkv, 1, k.shape[2], D)[lo:hi].reshape(1, hi - lo, k.shape[2], D)
    vf = v.reshape(B * Hkv, 1, v.shape[2], D)[lo:hi].reshape(1, hi - lo, v.shape[2], D)
    return qf, kf, vf, (lo, hi)
