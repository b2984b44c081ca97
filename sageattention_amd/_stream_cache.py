"""Zero-on-entry counter blocks, one per (purpose, device, stream).

Two kernels of the library synchronise through small blocks of device memory that must be ZERO when a launch starts and that the launch
itself returns to zero before it ends: the per-head arrival counters of the one-launch K / V pre-pass (``sage_prepass_kv``'s ``sync``) and
the ticket counters of a persistent attention launch (``SageLaunchAttr.launch_ws``).  Launches on one stream run in order, so ONE block per
stream, zeroed once when it is created, serves every call on that stream without a memset launch per call; launches on different streams get
different blocks.  The blocks are this package's, not the library's: the C ABI keeps no state.

The contract has one weak point, a launch that does not run to its end (a failed launch, an exception between handing the block over and
the launch): its block may be left dirty, and the next launch on it would skip work items or pass a barrier early.  ``drop`` forgets a
block; every caller drops its block when the C call it was handed to returns an error, and the pre-pass guard drops a device's blocks when
it trips.  All access is under a lock (streams come and go from any thread)."""
from __future__ import annotations

import threading
from typing import Dict, Tuple

import torch

_LOCK = threading.Lock()
_CACHE: Dict[Tuple[str, int, int], torch.Tensor] = {}
_MAX_ENTRIES = 128


def _key(purpose: str, device: torch.device) -> Tuple[str, int, int]:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return purpose, idx, torch.cuda.current_stream(idx).cuda_stream


def zeroed(purpose: str, words: int, device: torch.device, min_words: int = 0) -> torch.Tensor:
    """The int32 block of ``purpose`` for the current stream of ``device``: at least ``words`` long, zero when first handed out and -- by the
    kernels' contract -- zero again after every launch that used it.  Inside a graph capture a fresh zeroed tensor is recorded with the
    capture instead (a captured graph must not depend on memory the cache may replace)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros((words,), dtype=torch.int32, device=device)
    key = _key(purpose, device)
    with _LOCK:
        buf = _CACHE.get(key)
        if buf is None or buf.numel() < words:
            if buf is None and len(_CACHE) >= _MAX_ENTRIES:       # the oldest entry leaves (its memory goes back to the caching allocator in the
                _CACHE.pop(next(iter(_CACHE)))                    # order of the stream it was allocated on, i.e. behind its last launch)
            buf = _CACHE[key] = torch.zeros((max(words, min_words),), dtype=torch.int32, device=device)
        return buf


def drop(purpose: str, device: torch.device) -> None:
    """Forget the current stream's block of ``purpose`` (a call it was handed to failed: it may not be zero)."""
    if torch.cuda.is_current_stream_capturing():
        return
    key = _key(purpose, device)
    with _LOCK:
        _CACHE.pop(key, None)


def drop_device(device: torch.device) -> None:
    """Forget every block of ``device`` (the pre-pass guard tripped: a give-up leaves its flag word set)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _LOCK:
        for key in [k for k in _CACHE if k[1] == idx]:
            _CACHE.pop(key)
