"""Host wrappers of the gfx950 quantisation kernels.

Mirrors the reference's ``sageattention/quant.py`` (``per_block_int8`` :22, ``per_warp_int8``
:105, ``per_channel_fp8`` :224) and the host halves of ``sageattention/triton/quant_per_block.py``
:49, ``quant_per_block_varlen.py`` :60, ``quant_per_thread.py`` :154 -- same names, same argument
meaning, same return shapes for the INT8 tensors and scales.  Like the reference, these
functions allocate the outputs and the native op only writes into them.

The one deliberate difference is V: the reference returns a transposed ``[B,H,D,ceil64(L)]``
tensor whose token order is permuted for NVIDIA's ``mma.m16n8k32`` (quant.py:233,
fused.cu:287-291).  That layout is private to (its quantiser, its kernel); ours is too: V is
returned as the gfx950 *tile image* ``[B, H, ceil(L/64), D, 64]`` documented in
``csrc/sage_prep_v.hip``.
"""
from __future__ import annotations

import ctypes
import os
import warnings
from typing import NamedTuple, Optional, Tuple

import torch

from . import _cabi, _stream_cache

LOG2E = 1.44269504  # literal used by the reference (quant_per_block.py:87)


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return _cabi.DTYPE_F16
    if t.dtype == torch.bfloat16:
        return _cabi.DTYPE_BF16
    raise AssertionError("Input tensors must be in dtype of torch.float16 or torch.bfloat16")


def _stream(t: torch.Tensor) -> int:
    """Raw hipStream_t of torch's current stream on the tensor's device."""
    return torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None else torch.cuda.current_device())


# The wrappers below hand raw pointers to the C ABI; under torch.compile they must run eagerly (the
# reference supports torch.compile in non-fullgraph mode only, README.md:30), so dynamo is told not to
# trace into them.  The fused attention itself is a registered custom op (ops.py) and IS traceable.
_eager = torch.compiler.disable


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dims(t: torch.Tensor, tensor_layout: str):
    """(B, H, L, D, stride_b, stride_h, stride_l) of a 4-D tensor in HND / NHD layout."""
    if tensor_layout == "HND":
        B, H, L, D = t.shape
        return B, H, L, D, t.stride(0), t.stride(1), t.stride(2)
    if tensor_layout == "NHD":
        B, L, H, D = t.shape
        return B, H, L, D, t.stride(0), t.stride(2), t.stride(1)
    raise ValueError(f"Unknown tensor layout: {tensor_layout}")


def _aligned(t: torch.Tensor, elems: int) -> torch.Tensor:
    """The kernels use 16-byte vector accesses; re-pack the rare tensor that is not aligned."""
    ok = t.data_ptr() % 16 == 0 and t.stride(-1) == 1 and all(s % elems == 0 for s in t.stride()[:-1])
    return t if ok else t.contiguous()


def _squeeze_km(km: Optional[torch.Tensor], tensor_layout: str) -> Optional[torch.Tensor]:
    if km is None:
        return None
    if km.dim() == 4:                      # keepdim mean: [B,H,1,D] (HND) / [B,1,H,D] (NHD)
        km = km.squeeze(2 if tensor_layout == "HND" else 1)
    return km.contiguous()


@_eager
def _quant(x, km, blk, warp, gran, is_key, style, pre_scale, tensor_layout, nslots):
    x = _aligned(x, 8)
    B, H, L, D, sb, sh, sl = _dims(x, tensor_layout)
    # INT8 rows are stored head-major ([B, H, L, D] in memory) whatever the logical layout: the attention kernel
    # streams 64-row K tiles, and rows H*D bytes apart land on a fraction of the L2 channels (measured 3.5x slower)
    out = torch.empty((B, H, L, D), dtype=torch.int8, device=x.device)
    if tensor_layout == "NHD":
        out = out.permute(0, 2, 1, 3)
    _, _, _, _, ob, oh, ol = _dims(out, tensor_layout)
    scale = torch.empty((B, H, ((L + blk - 1) // blk) * nslots), dtype=torch.float32, device=x.device)
    if km is not None:
        assert km.dtype == x.dtype and km.shape == (B, H, D), "km must be [B, H, D] in the dtype of k"
    rc = _cabi.load().sage_quant_qk_int8(
        _p(x), _p(km), _p(out), _p(scale), B, H, L, D, sb, sh, sl, ob, oh, ol,
        (H * D) if km is not None else 0, D if km is not None else 0,
        blk, warp, gran, int(is_key), style, float(pre_scale), _dtype_code(x), _stream(x))
    _cabi.check(rc, "sage_quant_qk_int8")
    return out, scale


def per_block_int8(q, k, km=None, BLKQ: int = 128, BLKK: int = 64, sm_scale: Optional[float] = None,
                   tensor_layout: str = "HND", quantization_backend: str = "triton", k_done=None):
    """Per-block INT8 quantisation of q (128 rows) and k (64 rows); ``sm_scale*log2e`` is folded
    into q.  Returns ``q_int8, q_scale[B,Hq,ceil(Lq/BLKQ)], k_int8, k_scale[B,Hkv,ceil(Lk/BLKK)]``.
    ``quantization_backend`` selects the reference's rounding convention: "triton"
    (quant_per_block.py:21-47) or "cuda" (quant.py:22-103 -> fused.cu:64-198)."""
    D = k.size(-1)
    if sm_scale is None:
        sm_scale = D ** -0.5
    style = {"triton": _cabi.QSTYLE_TRITON, "cuda": _cabi.QSTYLE_CUDA}[quantization_backend]
    km = _squeeze_km(km, tensor_layout)
    q_int8 = q_scale = None        # q=None: the K half only (the attention kernel quantises Q itself, sage_attn_fused_qblock_pv_f16)
    if q is not None:
        q_int8, q_scale = _quant(q, None, BLKQ, BLKQ, _cabi.GRAN_PER_BLOCK, False, style, sm_scale * LOG2E, tensor_layout, 1)
    if k_done is not None:          # (k_int8, k_scale) already produced by the one-launch pre-pass (core.sageattn_qk_int8_pv_fp16_triton)
        k_int8, k_scale = k_done
    else:
        k_int8, k_scale = _quant(k, km, BLKK, BLKK, _cabi.GRAN_PER_BLOCK, True, style, 1.0, tensor_layout, 1)
    return q_int8, q_scale, k_int8, k_scale


def per_warp_int8(q, k, km=None, BLKQ: int = 128, WARPQ: int = 32, BLKK: int = 64, tensor_layout: str = "HND"):
    """q: one scale per WARPQ rows inside each BLKQ block; k: one per BLKK rows with the mean
    subtraction fused; sm_scale is NOT folded (quant.py:105-180)."""
    km = _squeeze_km(km, tensor_layout)
    q_int8, q_scale = _quant(q, None, BLKQ, WARPQ, _cabi.GRAN_PER_WARP, False, _cabi.QSTYLE_CUDA, 1.0, tensor_layout, BLKQ // WARPQ)
    k_int8, k_scale = _quant(k, km, BLKK, BLKK, _cabi.GRAN_PER_BLOCK, True, _cabi.QSTYLE_CUDA, 1.0, tensor_layout, 1)
    return q_int8, q_scale, k_int8, k_scale


def per_thread_int8(q, k, km=None, BLKQ: int = 128, WARPQ: int = 32, BLKK: int = 64, WARPK: int = 64,
                    sm_scale: Optional[float] = None, tensor_layout: str = "HND"):
    """"per-thread" granularity (quant_per_thread.py:154-203): 8 q scales per WARPQ rows
    (rows r, r+8, r+16, .. share), 4 k scales per WARPK tokens (tokens 8i+2t, 8i+2t+1 share)."""
    km = _squeeze_km(km, tensor_layout)
    q_int8, q_scale = _quant(q, None, BLKQ, WARPQ, _cabi.GRAN_PER_THREAD, False, _cabi.QSTYLE_TRITON_THREAD, 1.0,
                             tensor_layout, (BLKQ // WARPQ) * 8)
    k_int8, k_scale = _quant(k, km, BLKK, WARPK, _cabi.GRAN_PER_THREAD, True, _cabi.QSTYLE_TRITON_THREAD, 1.0,
                             tensor_layout, (BLKK // WARPK) * 4)
    return q_int8, q_scale, k_int8, k_scale


def _cu_blocks(cu: torch.Tensor, blk: int) -> torch.Tensor:
    lens = cu[1:] - cu[:-1]
    return torch.nn.functional.pad(torch.cumsum((lens + blk - 1) // blk, dim=0), (1, 0), value=0).to(torch.int32)


class VarlenPlan(NamedTuple):
    """Index arrays of one packed-batch call (``sage_varlen_plan``), all on the device, none read by the host."""
    cu_qs: Optional[torch.Tensor]      # [nseq + 1] prefix sums of ceil(Lq_i / 128) (only when asked for)
    cu_ks: torch.Tensor                # [nseq + 1] prefix sums of ceil(Lk_i / 64): k scale blocks and V tiles
    order: torch.Tensor                # [nseq] sequences by descending query length (launches without a work list)
    items: Optional[torch.Tensor]      # [items_bound, 2] (sequence, query block), heaviest first -- the attention launch's work list
    hdr: Optional[torch.Tensor]        # [8] number of items, launch plan (group, fold, left), number of slabs, max Lk, sum Lk
    slab_first: Optional[torch.Tensor]  # [nseq + 3] prefix sums of the slab counts: the sequences, then the rows behind the last / before the first one
    slab_seq: Optional[torch.Tensor]   # [slab_bound] slab -> segment (K / V pre-pass; nseq / nseq + 1 = the gap rows, which only the K mean reads)
    items_bound: int                   # host-known bound of the item count: ceil(sum Lq / 128) + nseq
    slab_bound: int                    # host-known bound of the slab count: ceil(rows of k / 512) + nseq + 2


@_eager
def varlen_plan(cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor, BLKQ: int = 128, BLKK: int = 64, want_q_blocks: bool = False,
                total_q: Optional[int] = None, total_k: Optional[int] = None, is_causal: bool = False, Hq: int = 1, Hkv: int = 1,
                head_dim: int = 128, pv_fp8: bool = False) -> Optional[VarlenPlan]:
    """Every index array of a packed-batch call in one launch (``sage_varlen_plan``): the block-count prefix sums (the reference's
    torch ops, quant_per_block_varlen.py:68-73), and -- when the packed token counts ``total_q`` / ``total_k`` (``q.shape[0]``,
    ``k.shape[0]``: known on the host) are given -- the attention launch's work list with its plan and the slab map of the one-launch
    K / V pre-pass.  ``cu_seqlens_*`` int32, contiguous; at most ``sage_varlen_plan_max_seqs()`` sequences (``None`` otherwise)."""
    nseq = cu_seqlens_q.shape[0] - 1
    lib = _cabi.load()
    if nseq < 1 or nseq > lib.sage_varlen_plan_max_seqs() or cu_seqlens_q.dtype != torch.int32 or cu_seqlens_k.dtype != torch.int32:
        return None
    dev = cu_seqlens_q.device
    cu_qs = torch.empty((nseq + 1,), dtype=torch.int32, device=dev) if want_q_blocks else None
    cu_ks = torch.empty((nseq + 1,), dtype=torch.int32, device=dev)
    order = torch.empty((nseq,), dtype=torch.int32, device=dev)
    items = hdr = slab_first = slab_seq = None
    items_bound = slab_bound = 0
    work = total_q is not None and total_k is not None and BLKQ == 128 and BLKK == 64
    if work:
        items_bound = (int(total_q) + 127) // 128 + nseq
        slab_bound = (int(total_k) + 511) // 512 + nseq + 2          # (+ 2: the gap segments outside every sequence, see the header)
        items = torch.empty((items_bound, 2), dtype=torch.int32, device=dev)
        hdr = torch.empty((8,), dtype=torch.int32, device=dev)
        slab_first = torch.empty((nseq + 3,), dtype=torch.int32, device=dev)
        slab_seq = torch.empty((slab_bound,), dtype=torch.int32, device=dev)
    rc = lib.sage_varlen_plan(_p(cu_seqlens_q), _p(cu_seqlens_k), nseq, int(total_k) if work else 0, BLKQ, BLKK, int(is_causal), int(Hq), int(Hkv), int(head_dim),
                              int(pv_fp8), _p(cu_qs), _p(cu_ks), _p(order), _p(items), items_bound, _p(slab_first), _p(slab_seq), slab_bound, _p(hdr),
                              _stream(cu_seqlens_q))
    _cabi.check(rc, "sage_varlen_plan")
    return VarlenPlan(cu_qs, cu_ks, order, items, hdr, slab_first, slab_seq, items_bound, slab_bound)


def per_block_int8_varlen(q, k, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, km=None,
                          BLKQ: int = 128, BLKK: int = 64, sm_scale: Optional[float] = None, cu_ks: Optional[torch.Tensor] = None,
                          cu_qs: Optional[torch.Tensor] = None):
    """Packed ``[sum L, H, D]`` per-block quantisation (quant_per_block_varlen.py:60-104).
    ``km`` (``[1, H, D]`` or ``[H, D]``) is subtracted from k inside the kernel, rounded to the
    input dtype exactly as the reference's ``k = k - km`` (core.py:432-434) does.  ``q`` or ``k`` may be None (one half only).
    With the prefix arrays of a ``varlen_plan`` (``cu_qs`` / ``cu_ks``) the scale tensors are allocated at their host-known bounds
    ``ceil(sum L / BLK) + nseq`` and nothing synchronises; without them they have the reference's exact shapes
    (``cu_seqlens_*_scale[-1]`` rows: one host synchronisation each, as quant_per_block_varlen.py:75-76)."""
    some = q if q is not None else k
    D = some.shape[-1]
    if sm_scale is None:
        sm_scale = D ** -0.5
    nseq = cu_seqlens_k.shape[0] - 1
    cu_q = cu_seqlens_q.to(torch.int32).contiguous()
    cu_k = cu_seqlens_k.to(torch.int32).contiguous()
    lib = _cabi.load()
    q_int8 = q_scale = k_int8 = k_scale = None
    if q is not None:
        q = _aligned(q, 8)
        Hq = q.shape[1]
        # head-major storage behind the packed [sum L, H, D] view (see _quant)
        q_int8 = torch.empty((Hq, q.shape[0], D), dtype=torch.int8, device=q.device).permute(1, 0, 2)
        if cu_qs is None:
            cu_qs = _cu_blocks(cu_q, BLKQ)
            nq = int(cu_qs[-1].item())
        else:
            nq = (q.shape[0] + BLKQ - 1) // BLKQ + nseq
        q_scale = torch.empty((nq, Hq), dtype=torch.float32, device=q.device)
        rc = lib.sage_quant_qk_int8_varlen(_p(q), None, _p(q_int8), _p(q_scale), _p(cu_q), _p(cu_qs), nseq, int(max_seqlen_q),
                                           Hq, D, q.stride(0), q.stride(1), q_int8.stride(0), q_int8.stride(1), 0,
                                           BLKQ, float(sm_scale * LOG2E), _dtype_code(q), _stream(q))
        _cabi.check(rc, "sage_quant_qk_int8_varlen(q)")
    if k is not None:
        k = _aligned(k, 8)
        Hkv = k.shape[1]
        k_int8 = torch.empty((Hkv, k.shape[0], D), dtype=torch.int8, device=k.device).permute(1, 0, 2)
        if cu_ks is None:
            cu_ks = _cu_blocks(cu_k, BLKK)
            nk = int(cu_ks[-1].item())
        else:
            nk = (k.shape[0] + BLKK - 1) // BLKK + nseq
        k_scale = torch.empty((nk, Hkv), dtype=torch.float32, device=k.device)
        if km is not None:
            km = km.reshape(Hkv, D).contiguous()
        rc = lib.sage_quant_qk_int8_varlen(_p(k), _p(km), _p(k_int8), _p(k_scale), _p(cu_k), _p(cu_ks), nseq, int(max_seqlen_k),
                                           Hkv, D, k.stride(0), k.stride(1), k_int8.stride(0), k_int8.stride(1), D,
                                           BLKK, 1.0, _dtype_code(k), _stream(k))
        _cabi.check(rc, "sage_quant_qk_int8_varlen(k)")
    return q_int8, q_scale, k_int8, k_scale, cu_qs, cu_ks


def _stats_ws(B: int, H: int, L: int, D: int, device) -> torch.Tensor:
    return torch.empty((int(_cabi.load().sage_stats_ws_floats(B, H, L, D)),), dtype=torch.float32, device=device)


@_eager
def channel_mean(x: torch.Tensor, tensor_layout: str = "HND") -> torch.Tensor:
    """``x.mean(dim=seq)`` in the input dtype, shape ``[B, H, D]`` -- the K-smoothing mean
    (core.py:280) and the V mean of ``sub_mean`` (quant.py:216) as one deterministic HIP reduction
    (fp32 accumulation, one rounding) instead of a torch op."""
    x = _aligned(x, 8)
    B, H, L, D, sb, sh, sl = _dims(x, tensor_layout)
    out = torch.empty((B, H, D), dtype=x.dtype, device=x.device)
    ws = _stats_ws(B, H, L, D, x.device)
    rc = _cabi.load().sage_channel_mean(_p(x), _p(out), _p(ws), B, H, L, D, sb, sh, sl, _dtype_code(x), _stream(x))
    _cabi.check(rc, "sage_channel_mean")
    return out


@_eager
def channel_mean_packed(x: torch.Tensor, cu_seqlens: Optional[torch.Tensor] = None, plan: Optional["VarlenPlan"] = None) -> torch.Tensor:
    """Mean over ALL tokens of a packed ``[sum L, H, D]`` tensor -> ``[1, H, D]`` (core.py:432-434).  With the slab map of a
    ``varlen_plan`` the sum runs over the per-sequence 512-token slabs of the one-launch pre-pass (same bits as ``prepass_kv_varlen``);
    without one over the packed tokens in slabs of 512."""
    x = _aligned(x, 8)
    T, H, D = x.shape
    out = torch.empty((1, H, D), dtype=x.dtype, device=x.device)
    if plan is not None and plan.slab_seq is not None:
        ws = _stats_ws(1, H, 512 * plan.slab_bound, D, x.device)
        rc = _cabi.load().sage_channel_mean_varlen(_p(x), _p(out), _p(ws), _p(cu_seqlens), _p(plan.slab_first), _p(plan.slab_seq), _p(plan.hdr),
                                                   cu_seqlens.shape[0] - 1, T, plan.slab_bound, H, D, x.stride(0), x.stride(1), _dtype_code(x), _stream(x))
        _cabi.check(rc, "sage_channel_mean_varlen")
        return out
    ws = _stats_ws(1, H, T, D, x.device)
    rc = _cabi.load().sage_channel_mean(_p(x), _p(out), _p(ws), 1, H, T, D, 0, x.stride(1), x.stride(0), _dtype_code(x), _stream(x))
    _cabi.check(rc, "sage_channel_mean")
    return out


_DEBUG = bool(int(os.environ.get("SAGE_DEBUG", "0") or 0))   # SAGE_DEBUG=1: check the pre-pass give-up flags after every call (synchronises)


@_eager
def per_channel_fp8(v: torch.Tensor, tensor_layout: str = "HND", scale_max: float = 448.0, smooth_v: bool = False
                    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """Per-channel FP8 (e4m3fn) quantisation of V fused with the transpose into the PV tile image
    (reference: quant.py:224-293).  Returns ``(v_image uint8 [B,H,ceil(L/64),D,64],
    v_scale fp32 [B,H,D], vm)`` where ``vm`` is the fp32 per-channel mean ``[B,H,D]`` that was
    subtracted when ``smooth_v`` (to be added back by the attention epilogue), else None."""
    v = _aligned(v, 8)
    B, H, L, D, sb, sh, sl = _dims(v, tensor_layout)
    nt = (L + 63) // 64
    v_image = torch.empty((B, H, nt, D, 64), dtype=torch.uint8, device=v.device)
    v_scale = torch.empty((B, H, D), dtype=torch.float32, device=v.device)
    vm = torch.empty((B, H, D), dtype=torch.float32, device=v.device) if smooth_v else None
    ws = _stats_ws(B, H, L, D, v.device)
    rc = _cabi.load().sage_prep_v_fp8(_p(v), _p(v_image), _p(v_scale), _p(vm), _p(ws), B, H, L, D, sb, sh, sl,
                                      float(scale_max), _dtype_code(v), _stream(v))
    _cabi.check(rc, "sage_prep_v_fp8")
    return v_image, v_scale, vm


def _prepass_sync(B: int, H: int, device) -> torch.Tensor:
    """The per-head counters of the fused pre-pass (``sync`` of ``sage_prepass_kv``): ZERO on entry, and returned to zero by the kernel
    before it ends, so one block per (device, stream), zeroed ONCE when it is created, serves every call issued on that stream -- no
    zeroing launch per call (``_stream_cache``: locked, dropped when a call fails or the guard trips)."""
    return _stream_cache.zeroed("prepass", int(_cabi.load().sage_prepass_sync_words(B, H)), device, min_words=4096)


def prepass_failed_heads(sync: torch.Tensor, B: int, H: int) -> int:
    """Synchronise and count the (K|V, batch, head) entries of a pre-pass call in which a workgroup stopped waiting for its head-mates and
    computed the head's statistics itself (the outputs are right; 0 = nobody had to).  Debugging / test aid: ``SAGE_DEBUG=1`` makes
    ``prepass_kv_fp8`` call it and warn."""
    n = int(_cabi.load().sage_prepass_failed_heads(_p(sync), B, H, _stream(sync)))
    if n < 0:
        _cabi.check(n, "sage_prepass_failed_heads")
    return n


class _PrepassGuard:
    """Per-device watch on the one-launch pre-pass's in-launch head barrier.  The barrier wants every slab of a head resident at once; the
    C ABI bounds that by the compute units the launch stream may use, but it cannot see compute units that OTHER streams' kernels hold
    (an RCCL kernel beside the attention stream).  A workgroup that waits in vain (tens of milliseconds) computes the head's statistics
    itself -- the result is still right, the launch is slow -- and stores 1 into this guard's pinned host word (``host_flag`` of
    ``sage_prepass_kv``).  The word is read -- a plain host memory read, no synchronisation -- at the start of every later pre-pass on the
    device: once it is set the device's calls take the kernel sequence for ``REARM_SECONDS`` (doubling with every further trip; a warning
    says so once) -- the stall that tripped it may have been transient -- and then try the one launch again."""
    _by_device: dict = {}

    def __init__(self, device: torch.device):
        with torch.cuda.device(device):
            host, dev = ctypes.c_void_p(), ctypes.c_void_p()
            _cabi.check(_cabi.load().sage_host_word_alloc(ctypes.byref(host), ctypes.byref(dev)), "sage_host_word_alloc")
        self.host, self.ptr = host.value, dev.value          # lives as long as the process (one 64-byte pinned block per device)
        self.tripped = False
        self.device = device

    @classmethod
    def of(cls, device: torch.device) -> "_PrepassGuard":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        g = cls._by_device.get(idx)
        if g is None:
            g = cls._by_device[idx] = cls(torch.device("cuda", idx))
        return g

    REARM_SECONDS = 30.0        # a tripped guard lets the one-launch route try again after this long (a transient stall must not cost the process)

    def fused_allowed(self) -> bool:
        import time
        if not self.tripped and ctypes.c_int32.from_address(self.host).value != 0:
            self.tripped = True
            _stream_cache.drop_device(self.device)      # (a give-up leaves its flag word set in the head's sync line)
            self.tripped_at = time.monotonic()
            self.trips = getattr(self, "trips", 0) + 1
            if self.trips == 1:
                warnings.warn(f"sageattention_amd: a one-launch K/V pre-pass on {self.device} stopped waiting for the other slabs of a head "
                              "(compute units held by another stream?) and recomputed the head's statistics per workgroup: that call was "
                              f"correct but slow.  This device's calls take the kernel sequence for the next {self.REARM_SECONDS:.0f} s.",
                              RuntimeWarning, stacklevel=3)
        elif self.tripped and time.monotonic() - self.tripped_at > self.REARM_SECONDS * min(2 ** (self.trips - 1), 64):
            # (every further trip doubles the pause: a device that really cannot hold a head's slabs settles on the kernel sequence)
            ctypes.c_int32.from_address(self.host).value = 0
            self.tripped = False
        return not self.tripped

    def reset(self) -> None:          # tests
        ctypes.c_int32.from_address(self.host).value = 0
        self.tripped = False
        self.trips = 0


def prepass_fused_ok(k: torch.Tensor, tensor_layout: str = "HND") -> bool:
    """Whether the one-launch pre-pass covers this K / V length on the current stream (the slabs of a head wait for each other in the
    launch: one per compute unit the stream may use, ``sage_prepass_max_seqlen_stream``) -- and has not failed on this device before."""
    _, _, L, D, _, _, sl = _dims(k, tensor_layout)
    # the kernel addresses one head with 32-bit buffer offsets (row stride x rows x 2 bytes)
    # (rows of the last 512-row slab past L go through the buffer range check: their offsets must not wrap either)
    lpad = (L + 511) // 512 * 512
    if ((lpad - 1) * sl + D) * 2 >= 2 ** 32 or not k.is_cuda:
        return False
    return L <= int(_cabi.load().sage_prepass_max_seqlen_stream(_stream(k))) and _PrepassGuard.of(k.device).fused_allowed()


@_eager
def prepass_kv_fp8(k: torch.Tensor, v: Optional[torch.Tensor], tensor_layout: str = "HND", smooth_k: bool = True,
                   smooth_v: bool = False, BLKK: int = 64, qk_quant_gran: str = "per_thread", scale_max: float = 448.0,
                   v_fp16: bool = False, sync: Optional[torch.Tensor] = None):
    """K and V pre-pass of the FP8-PV entry points in ONE launch that reads K and V once: the bits of
    ``channel_mean`` + ``per_thread_int8`` / ``per_warp_int8`` (K side) + ``per_channel_fp8``.
    Returns ``(km [B,H,D] | None, k_int8, k_scale, v_image, v_scale, vm)``; ``v=None`` runs the K half only
    (``v_image, v_scale, vm`` are None).  ``qk_quant_gran`` "per_thread" gives 4 k scales per BLKK keys with the
    Triton-per-thread rounding, "per_warp" / "per_block" one scale per BLKK keys with the CUDA rounding
    (quant.py:105-180) -- the K conventions of the reference's CUDA entry points --, "per_block_triton" one scale per BLKK keys with
    the Triton rounding (quant_per_block.py:21-46, the Triton-named API).  ``v_fp16=True`` (FP16-PV entry points) makes
    the V half the fp16 tile image of ``prep_v_fp16`` instead (``v_scale`` and ``vm`` are then None).  ``sync``: optional
    caller-owned int32 buffer of ``sage_prepass_sync_words(B, H)`` words, ZERO on entry (the kernel leaves its counters at zero; to
    inspect with ``prepass_failed_heads`` afterwards)."""
    k = _aligned(k, 8)
    B, H, L, D, k_sb, k_sh, k_sl = _dims(k, tensor_layout)
    dev = k.device
    k_int8 = torch.empty((B, H, L, D), dtype=torch.int8, device=dev)          # head-major in memory (see _quant)
    if tensor_layout == "NHD":
        k_int8 = k_int8.permute(0, 2, 1, 3)
    _, _, _, _, ob, oh, ol = _dims(k_int8, tensor_layout)
    if qk_quant_gran == "per_thread":
        gran, style, slots = _cabi.GRAN_PER_THREAD, _cabi.QSTYLE_TRITON_THREAD, 4
    elif qk_quant_gran == "per_block_triton":      # the Triton-named API's K half: per-block scales, Triton rounding (quant_per_block.py:21-46)
        gran, style, slots = _cabi.GRAN_PER_BLOCK, _cabi.QSTYLE_TRITON, 1
    else:
        gran, style, slots = _cabi.GRAN_PER_BLOCK, _cabi.QSTYLE_CUDA, 1
    k_scale = torch.empty((B, H, ((L + BLKK - 1) // BLKK) * slots), dtype=torch.float32, device=dev)
    km = torch.empty((B, H, D), dtype=k.dtype, device=dev) if smooth_k else None
    v_image = v_scale = vm = None
    v_sb = v_sh = v_sl = 0
    if v is not None:
        v = _aligned(v, 8)
        assert _dims(v, tensor_layout)[:4] == (B, H, L, D) and v.dtype == k.dtype, "k and v must have one shape and dtype"
        _, _, _, _, v_sb, v_sh, v_sl = _dims(v, tensor_layout)
        if v_fp16:
            assert not smooth_v, "the fp16 image has no smooth_v here (sub_mean goes through prep_v_fp16)"
            v_image = torch.empty((B, H, (L + 63) // 64, D, 64), dtype=torch.float16, device=dev)
        else:
            v_image = torch.empty((B, H, (L + 63) // 64, D, 64), dtype=torch.uint8, device=dev)
            v_scale = torch.empty((B, H, D), dtype=torch.float32, device=dev)
            vm = torch.empty((B, H, D), dtype=torch.float32, device=dev) if smooth_v else None
    lib = _cabi.load()
    ws = torch.empty((int(lib.sage_prepass_ws_floats(B, H, L, D)),), dtype=torch.float32, device=dev)
    if sync is None:
        sync = _prepass_sync(B, H, dev)
    assert sync.dtype == torch.int32 and sync.numel() >= int(lib.sage_prepass_sync_words(B, H)) and sync.device == dev
    rc = lib.sage_prepass_kv(_p(k), _p(v), _p(km), _p(k_int8), _p(k_scale), _p(v_image), _p(v_scale), _p(vm), _p(ws), _p(sync),
                             B, H, L, D, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, ob, oh, ol,
                             BLKK, gran, style, float(scale_max), int(bool(v_fp16)), _dtype_code(k), _PrepassGuard.of(dev).ptr, _stream(k))
    if rc != 0:
        _stream_cache.drop("prepass", dev)         # (the block may not be zero any more)
    _cabi.check(rc, "sage_prepass_kv")
    if _DEBUG:
        n = prepass_failed_heads(sync, B, H)
        if n:
            sync.zero_()           # (the give-up flags, word 2 of a head's line, are the one thing the kernel does not clear itself)
            warnings.warn(f"sage_prepass_kv: in {n} (K|V, batch, head) entries a workgroup stopped waiting for the other slabs of its head and "
                          "recomputed the statistics itself (slow, not wrong); is the stream restricted to few compute units?", RuntimeWarning)
    return km, k_int8, k_scale, v_image, v_scale, vm


def prepass_varlen_fused_ok(k: torch.Tensor, plan: Optional["VarlenPlan"], max_seqlen_k: int, smooth_k: bool = True) -> bool:
    """Whether ``prepass_kv_varlen`` takes this packed batch: a slab map from ``varlen_plan``; with ``smooth_k`` every slab of a head --
    all sequences -- waits for the others inside the launch, so the host-known bound of their number must not exceed 128 nor the
    compute units of the stream; 32-bit offsets inside one sequence of one head."""
    if plan is None or plan.slab_seq is None or not k.is_cuda:
        return False
    lpad = (int(max_seqlen_k) + 511) // 512 * 512
    if ((lpad - 1) * k.stride(0) + k.shape[-1]) * 2 >= 2 ** 32:
        return False
    if not smooth_k:
        return True
    return 512 * plan.slab_bound <= int(_cabi.load().sage_prepass_max_seqlen_stream(_stream(k))) and _PrepassGuard.of(k.device).fused_allowed()


@_eager
def prepass_kv_varlen(k: torch.Tensor, v: Optional[torch.Tensor], cu_seqlens_k: torch.Tensor, plan: "VarlenPlan", max_seqlen_k: int,
                      smooth_k: bool = True, sync: Optional[torch.Tensor] = None):
    """The K / V pre-pass of ``sageattn_varlen`` (core.py:431-444) in ONE launch that reads K and V once (``sage_prepass_kv_varlen``):
    ``km = k.mean(dim=0)`` over all packed tokens, per-sequence INT8 ``k - km`` with one scale per 64 keys (Triton rounding), the fp16 V
    tile image.  Returns ``(km [1, H, D] | None, k_int8 [sum L, H, D] (head-major storage), k_scale [nblk_bound, H], v_image | None)`` --
    the bits of ``channel_mean_packed(plan)`` + ``per_block_int8_varlen`` (K half) + ``prep_v_fp16_varlen``."""
    k = _aligned(k, 8)
    T, H, D = k.shape
    dev = k.device
    nseq = cu_seqlens_k.shape[0] - 1
    k_int8 = torch.empty((H, T, D), dtype=torch.int8, device=dev).permute(1, 0, 2)
    nblk = (T + 63) // 64 + nseq
    k_scale = torch.empty((nblk, H), dtype=torch.float32, device=dev)
    km = torch.empty((1, H, D), dtype=k.dtype, device=dev) if smooth_k else None
    v_image = None
    v_sl = v_sh = 0
    if v is not None:
        v = _aligned(v, 8)
        assert v.shape == k.shape and v.dtype == k.dtype, "k and v must have one shape and dtype"
        v_sl, v_sh = v.stride(0), v.stride(1)
        v_image = torch.empty((nblk, H, D, 64), dtype=torch.float16, device=dev)
    lib = _cabi.load()
    ws = torch.empty((int(lib.sage_prepass_ws_floats(1, H, 512 * plan.slab_bound, D)),), dtype=torch.float32, device=dev)
    if sync is None:
        sync = _prepass_sync(1, H, dev)
    rc = lib.sage_prepass_kv_varlen(_p(k), _p(v), _p(km), _p(k_int8), _p(k_scale), _p(v_image), _p(ws), _p(sync), _p(cu_seqlens_k),
                                    _p(plan.cu_ks), _p(plan.slab_first), _p(plan.slab_seq), _p(plan.hdr), nseq, T, int(max_seqlen_k),
                                    plan.slab_bound, H, D, k.stride(0), k.stride(1), v_sl, v_sh, k_int8.stride(0), k_int8.stride(1),
                                    _dtype_code(k), _PrepassGuard.of(dev).ptr, _stream(k))
    if rc != 0:
        _stream_cache.drop("prepass", dev)
    _cabi.check(rc, "sage_prepass_kv_varlen")
    if _DEBUG:
        n = prepass_failed_heads(sync, 1, H)
        if n:
            sync.zero_()
            warnings.warn(f"sage_prepass_kv_varlen: in {n} heads a workgroup stopped waiting for the other slabs and recomputed the statistics "
                          "itself (slow, not wrong)", RuntimeWarning)
    return km, k_int8, k_scale, v_image


@_eager
def prep_v_fp16(v: torch.Tensor, tensor_layout: str = "HND", vm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FP16-PV paths: ``v.to(float16)`` (core.py:297-298,613) -- or ``(v - vm).to(float16)`` when a
    per-channel mean ``vm`` fp32 ``[B,H,D]`` is given (``sub_mean``) -- fused with the transpose into
    the tile image ``[B,H,ceil(L/64),D,64]`` (fp16)."""
    v = _aligned(v, 8)
    B, H, L, D, sb, sh, sl = _dims(v, tensor_layout)
    v_image = torch.empty((B, H, (L + 63) // 64, D, 64), dtype=torch.float16, device=v.device)
    if vm is not None:
        assert vm.dtype == torch.float32 and vm.shape == (B, H, D) and vm.is_contiguous()
    rc = _cabi.load().sage_prep_v_f16(_p(v), _p(v_image), _p(vm), B, H, L, D, sb, sh, sl, _dtype_code(v), _stream(v))
    _cabi.check(rc, "sage_prep_v_f16")
    return v_image


def sub_mean(v: torch.Tensor, tensor_layout: str = "HND"):
    """Reference ``sub_mean`` (quant.py:182-222): returns ``(smoothed V as fp16 tile image, vm [B,H,D] in
    the dtype of v)``; the mean is added back in the attention epilogue."""
    vm = channel_mean(v, tensor_layout)
    return prep_v_fp16(v, tensor_layout, vm=vm.float()), vm


@_eager
def prep_v_fp16_varlen(v: torch.Tensor, cu_seqlens_k: torch.Tensor, cu_tiles: torch.Tensor, max_seqlen_k: int,
                       ntiles: Optional[int] = None) -> torch.Tensor:
    """Packed ``[sum L, H, D]`` V -> tile image ``[cu_tiles[-1], H, D, 64]`` fp16.  ``ntiles``: an upper bound of ``cu_tiles[-1]``
    known on the host (``ceil(sum L / 64) + nseq``) instead of the host synchronisation on the exact count."""
    v = _aligned(v, 8)
    H, D = v.shape[1], v.shape[2]
    if ntiles is None:
        ntiles = int(cu_tiles[-1].item())
    v_image = torch.empty((ntiles, H, D, 64), dtype=torch.float16, device=v.device)
    rc = _cabi.load().sage_prep_v_f16_varlen(_p(v), _p(v_image), _p(cu_seqlens_k), _p(cu_tiles), cu_seqlens_k.shape[0] - 1,
                                             int(max_seqlen_k), H, D, v.stride(0), v.stride(1), _dtype_code(v), _stream(v))
    _cabi.check(rc, "sage_prep_v_f16_varlen")
    return v_image
