"""HIP-graph replay of the dense ``sageattn()`` pipeline for launch-bound shapes.

One ``sageattn()`` call is two kernel launches (the one-launch K / V pre-pass, the attention kernel with the Q quantiser in its
prologue; three with ``return_lse``'s correction matmul) and about eight allocations: tens of microseconds of host work, more than the
GPU needs below N ~ 2k.  The dense pipeline has no host synchronisation and launches on the caller's stream, so it can
be captured once per (shape, dtype, flags) and replayed with one ``hipGraphLaunch``.  The caller writes the inputs
into the graph's static tensors (``.q .k .v``, e.g. as the output buffers of its projection GEMMs) or lets
``__call__`` copy them in.
"""
from __future__ import annotations

from typing import Optional

import torch

from .core import sageattn


class GraphedSageAttn:
    def __init__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_fn=None, **kwargs):
        """Capture ``attn_fn(q, k, v, **kwargs)`` (default :func:`sageattn`) for tensors shaped/laid out like q, k, v."""
        assert q.is_cuda, "HIP graphs need device tensors"
        self.attn_fn = attn_fn or sageattn
        self.kwargs = kwargs
        self.q, self.k, self.v = q.clone(), k.clone(), v.clone()
        side = torch.cuda.Stream(device=q.device)
        side.wait_stream(torch.cuda.current_stream(q.device))
        with torch.cuda.stream(side):                    # warm-up off the capture: library load, allocator pools
            for _ in range(2):
                self.attn_fn(self.q, self.k, self.v, **kwargs)
        torch.cuda.current_stream(q.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.attn_fn(self.q, self.k, self.v, **kwargs)

    def replay(self):
        """Inputs already written into ``.q .k .v``; returns the static output tensor(s)."""
        self.graph.replay()
        return self.out

    def __call__(self, q: Optional[torch.Tensor] = None, k: Optional[torch.Tensor] = None, v: Optional[torch.Tensor] = None):
        for dst, src in ((self.q, q), (self.k, k), (self.v, v)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        return self.replay()
