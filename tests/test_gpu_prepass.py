"""The one-launch K / V pre-pass (sage_prepass_kv) against the kernel sequence it replaces: every output bit-equal.

The sequence (sage_channel_mean + sage_quant_qk_int8 + sage_prep_v_fp8) is itself pinned to the oracle by
tests/test_gpu_parity.py, so bit-equality here carries that parity over."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from sageattention_amd import _cabi, quant


def _mk(B, H, L, D, dtype, layout, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    shape = (B, H, L, D) if layout == "HND" else (B, L, H, D)
    k = torch.randn(shape, device="cuda", dtype=torch.float32, generator=g)
    v = torch.randn(shape, device="cuda", dtype=torch.float32, generator=g)
    chan = torch.linspace(-2.0, 3.0, D, device="cuda")
    return (k * 1.5 + chan).to(dtype), (v * (1.0 + chan.abs()) + 0.5 * chan).to(dtype)


def _sequence(k, v, layout, smooth_k, smooth_v, blkk, gran):
    km = quant.channel_mean(k, layout) if smooth_k else None
    if gran == "per_thread":
        k8, ks = quant._quant(k, km, blkk, blkk, _cabi.GRAN_PER_THREAD, True, _cabi.QSTYLE_TRITON_THREAD, 1.0, layout, 4)
    elif gran == "per_block_triton":
        k8, ks = quant._quant(k, km, blkk, blkk, _cabi.GRAN_PER_BLOCK, True, _cabi.QSTYLE_TRITON, 1.0, layout, 1)
    else:
        k8, ks = quant._quant(k, km, blkk, blkk, _cabi.GRAN_PER_BLOCK, True, _cabi.QSTYLE_CUDA, 1.0, layout, 1)
    vi, vs, vm = quant.per_channel_fp8(v, tensor_layout=layout, scale_max=448.0, smooth_v=smooth_v)
    return km, k8, ks, vi, vs, vm


def _same(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    assert a.shape == b.shape and a.dtype == b.dtype, what
    if a.dtype in (torch.float16, torch.bfloat16):
        a, b = a.view(torch.int16), b.view(torch.int16)
    elif a.dtype == torch.float32:
        a, b = a.view(torch.int32), b.view(torch.int32)
    assert torch.equal(a, b), f"{what}: {(a != b).sum().item()} of {a.numel()} differ"


CASES = [  # B, H, L, D, dtype, layout, smooth_k, smooth_v, blkk, gran
    (2, 4, 1024, 128, torch.bfloat16, "HND", True, False, 64, "per_thread"),
    (1, 3, 4096, 128, torch.float16, "HND", True, True, 64, "per_thread"),
    (2, 2, 1000, 128, torch.bfloat16, "NHD", True, True, 64, "per_thread"),       # ragged tail, L % 16 != 0
    (1, 2, 77, 128, torch.float16, "HND", True, True, 64, "per_warp"),            # one partial slab
    (2, 3, 2048 + 513, 64, torch.bfloat16, "HND", True, False, 64, "per_thread"),
    (1, 5, 1536, 64, torch.float16, "NHD", True, True, 64, "per_warp"),
    (1, 2, 3000, 128, torch.bfloat16, "HND", True, False, 128, "per_thread"),     # the sm90 entry point's 128-key groups
    (2, 2, 640, 64, torch.float16, "HND", True, False, 128, "per_warp"),
    (1, 2, 2048, 128, torch.bfloat16, "HND", False, False, 64, "per_thread"),     # smooth_k off: no statistics for K
    (1, 1, 32768, 128, torch.bfloat16, "HND", True, True, 64, "per_thread"),      # 64 slabs
    (1, 2, 65536, 128, torch.float16, "HND", True, True, 64, "per_thread"),       # 128 slabs: the longest head the barrier takes
    (1, 1, 65536 - 300, 64, torch.bfloat16, "NHD", True, False, 64, "per_warp"),
    (1, 2, 1, 64, torch.float16, "HND", True, True, 64, "per_thread"),
    # the Triton-named API's K convention: per-block scales, Triton rounding (an all-zero block has scale 0)
    (2, 4, 1024, 128, torch.bfloat16, "HND", True, False, 64, "per_block_triton"),
    (1, 3, 777, 64, torch.float16, "NHD", False, False, 64, "per_block_triton"),
    (1, 2, 5000, 128, torch.float16, "NHD", True, True, 64, "per_block_triton"),
]


@pytest.mark.parametrize("B,H,L,D,dtype,layout,smooth_k,smooth_v,blkk,gran", CASES)
def test_fused_prepass_bit_equals_the_sequence(B, H, L, D, dtype, layout, smooth_k, smooth_v, blkk, gran):
    k, v = _mk(B, H, L, D, dtype, layout, 7 * L + D)
    ref = _sequence(k, v, layout, smooth_k, smooth_v, blkk, gran)
    sync = torch.full((int(_cabi.load().sage_prepass_sync_words(B, H)),), 0x5a5a5a5a, dtype=torch.int32, device="cuda")   # dirty on purpose
    for rep in range(3):                   # the same (dirty) sync buffer serves every call: the entry point zeroes it
        got = quant.prepass_kv_fp8(k, v, layout, smooth_k=smooth_k, smooth_v=smooth_v, BLKK=blkk, qk_quant_gran=gran, sync=sync)
        for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b, f"{name} (call {rep})")
    assert quant.prepass_failed_heads(sync, B, H) == 0
    assert int(sync.abs().sum().item()) == 0           # counters re-armed by the kernel, no give-up flag


def test_triton_api_one_launch_prepass_is_bit_identical():
    """sageattn_qk_int8_pv_fp16_triton through the one-launch pre-pass (K mean + Triton-rounded per-block INT8 K + fp16 V image) against the
    kernel sequence, incl. an all-zero K block (scale 0 -> INT8 zeros, as the stand-alone quantiser gives) and the masked kernels."""
    import sageattention_amd as sa
    g = torch.Generator(device="cuda").manual_seed(3)
    for (B, Hq, Hkv, L, D, dt, layout) in [(2, 8, 4, 1500, 128, torch.bfloat16, "HND"), (1, 4, 4, 640, 64, torch.float16, "NHD")]:
        shp = (lambda h: (B, h, L, D)) if layout == "HND" else (lambda h: (B, L, h, D))
        q = torch.randn(shp(Hq), device="cuda", generator=g).to(dt)
        k = (torch.randn(shp(Hkv), device="cuda", generator=g) + 0.7).to(dt)
        v = torch.randn(shp(Hkv), device="cuda", generator=g).to(dt)
        for smooth_k in (True, False):
            if not smooth_k:
                (k[:, :, 64:128] if layout == "HND" else k[:, 64:128]).zero_()
            for causal in (False, True):
                a, la = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, is_causal=causal, smooth_k=smooth_k, return_lse=True,
                                                           fused_prepass=True)
                b, lb = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, is_causal=causal, smooth_k=smooth_k, return_lse=True,
                                                           fused_prepass=False)
                assert torch.isfinite(a.float()).all()
                _same(a, b, "o"); _same(la, lb, "lse")
        mask = torch.rand(B, Hq, L, L, device="cuda", generator=g) > 0.3
        _same(sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, attn_mask=mask, fused_prepass=True),
              sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, attn_mask=mask, fused_prepass=False), "masked o")


def test_k_half_only_and_side_stream():
    k, v = _mk(2, 4, 2048, 128, torch.bfloat16, "HND", 3)
    ref = _sequence(k, v, "HND", True, False, 64, "per_thread")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = quant.prepass_kv_fp8(k, None, "HND", smooth_k=True)
    got_main = quant.prepass_kv_fp8(k, v, "HND", smooth_k=True)       # concurrently, on its own sync buffer
    torch.cuda.current_stream().wait_stream(side)
    for a, b, name in zip(got[:3], ref[:3], ("km", "k_int8", "k_scale")):
        _same(a, b, name + " (K half, side stream)")
    assert got[3] is None and got[4] is None and got[5] is None
    for a, b, name in zip(got_main, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
        _same(a, b, name)


def test_many_small_heads_and_full_chip():
    # more workgroups than the chip holds at once (2 * 64 * 16 * 2 = 4096), heads of 16 slabs each
    k, v = _mk(2, 64, 8192, 128, torch.bfloat16, "HND", 11)
    ref = _sequence(k, v, "HND", True, False, 64, "per_thread")
    got = quant.prepass_kv_fp8(k, v, "HND", smooth_k=True)
    for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
        _same(a, b, name)


def test_too_long_is_refused():
    k = torch.zeros(1, 1, 65536 + 512, 64, device="cuda", dtype=torch.float16)
    assert not quant.prepass_fused_ok(k)
    with pytest.raises(ValueError, match="too long"):
        quant.prepass_kv_fp8(k, k)


def test_graph_capture_replays():
    k, v = _mk(1, 4, 4096, 128, torch.bfloat16, "HND", 5)
    ref = _sequence(k, v, "HND", True, True, 64, "per_thread")
    quant.prepass_kv_fp8(k, v, "HND", smooth_k=True, smooth_v=True)     # warm the allocator outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = quant.prepass_kv_fp8(k, v, "HND", smooth_k=True, smooth_v=True)
    for _ in range(3):
        for t in got:
            if t is not None:
                t.zero_()
        g.replay()
        torch.cuda.synchronize()
        for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b, name + " (graph replay)")


ENTRY_CASES = [  # entry point, kwargs
    ("sageattn", dict()),
    ("sageattn_qk_int8_pv_fp8_cuda", dict(qk_quant_gran="per_warp", pv_accum_dtype="fp32+fp32")),
    ("sageattn_qk_int8_pv_fp8_cuda", dict(qk_quant_gran="per_thread", pv_accum_dtype="fp32", smooth_v=True)),
    ("sageattn_qk_int8_pv_fp8_cuda", dict(qk_quant_gran="per_block", pv_accum_dtype="fp32+fp16", smooth_k=False)),
    ("sageattn_qk_int8_pv_fp8_cuda_sm90", dict(qk_quant_gran="per_thread")),
    ("sageattn_qk_int8_pv_fp8_cuda_sm90", dict(qk_quant_gran="per_warp")),
    ("sageattn_qk_int8_pv_fp16_cuda", dict(qk_quant_gran="per_thread", pv_accum_dtype="fp32")),
    ("sageattn_qk_int8_pv_fp16_cuda", dict(qk_quant_gran="per_warp", pv_accum_dtype="fp16+fp32")),
]


@pytest.mark.parametrize("entry,extra", ENTRY_CASES)
@pytest.mark.parametrize("layout,causal,D", [("HND", True, 128), ("NHD", False, 64)])
def test_entry_points_are_bit_equal_with_either_prepass(entry, extra, layout, causal, D):
    """Every dense entry point gives the same bits whether its pre-pass is the one launch or the kernel sequence."""
    import sageattention_amd as sa
    g = torch.Generator(device="cuda").manual_seed(21)
    shape = (2, 4, 1500, D) if layout == "HND" else (2, 1500, 4, D)
    q, k, v = (torch.randn(shape, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
    k = k + 0.75
    fn = getattr(sa, entry)
    kw = dict(tensor_layout=layout, is_causal=causal, return_lse=True, **extra)
    o0, l0 = fn(q, k, v, fused_prepass=False, **kw)
    o1, l1 = fn(q, k, v, fused_prepass=True, **kw)
    o2, l2 = fn(q, k, v, **kw)                     # the default picks one of the two
    _same(o1, o0, "o")
    _same(l1, l0, "lse")
    _same(o2, o0, "o (default)")
    _same(l2, l0, "lse (default)")


def test_default_prepass_choice():
    from sageattention_amd import core
    mk = lambda *s: torch.empty(*s, device="cuda", dtype=torch.float16)
    assert core._fused_prepass_wanted(mk(2, 32, 8192, 128), "HND", None)
    assert core._fused_prepass_wanted(mk(1, 4, 200, 64), "HND", None)            # few heads: one launch instead of six
    assert not core._fused_prepass_wanted(mk(64, 16, 256, 64), "HND", None)      # many half-empty slabs
    assert not core._fused_prepass_wanted(mk(1, 2, 70000, 64), "HND", None)      # beyond the in-launch barrier's reach
    assert not core._fused_prepass_wanted(mk(1, 2, 70000, 64), "HND", True)
    assert not core._fused_prepass_wanted(mk(2, 32, 8192, 128), "HND", False)



def test_head_barrier_makes_progress_while_other_kernels_hold_the_chip():
    """The in-launch barrier needs every slab of the lowest unfinished head to get a slot.  Crowd the device: a long attention
    call and a second pre-pass on other streams while the pre-pass under test runs; it must finish, bit-equal, with its sync
    buffer back at zero (a workgroup that gave up waiting would leave its sticky flag)."""
    import sageattention_amd as sa
    k, v = _mk(2, 32, 8192, 128, torch.bfloat16, "HND", 17)
    ref = _sequence(k, v, "HND", True, False, 64, "per_thread")
    qa = torch.randn(1, 32, 16384, 128, device="cuda", dtype=torch.bfloat16)
    k2, v2 = _mk(1, 16, 32768, 128, torch.bfloat16, "HND", 23)
    ref2 = _sequence(k2, v2, "HND", True, True, 64, "per_thread")
    torch.cuda.synchronize()
    s_attn, s_pp = torch.cuda.Stream(), torch.cuda.Stream()
    words = int(_cabi.load().sage_prepass_sync_words(2, 32))
    syncs = [(torch.empty(words, dtype=torch.int32, device="cuda"), torch.empty(words, dtype=torch.int32, device="cuda")) for _ in range(4)]
    outs = []
    for rep in range(4):
        with torch.cuda.stream(s_attn):
            sa.sageattn(qa, qa, qa, is_causal=False)                      # ~5 ms of attention workgroups on every CU
        with torch.cuda.stream(s_pp):
            got2 = quant.prepass_kv_fp8(k2, v2, "HND", smooth_k=True, smooth_v=True, sync=syncs[rep][1])      # 64-slab heads
        outs.append((quant.prepass_kv_fp8(k, v, "HND", smooth_k=True, sync=syncs[rep][0]), got2))
    torch.cuda.synchronize()
    for got, got2 in outs:
        for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b, name)
        for a, b, name in zip(got2, ref2, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b, name + " (64-slab heads, side stream)")
    for s0, s1 in syncs:
        assert quant.prepass_failed_heads(s0, 2, 32) == 0 and quant.prepass_failed_heads(s1, 1, 16) == 0


def test_a_pre_pass_that_gives_up_is_loud():
    """The in-launch head barrier gives up after a bounded wait if the slabs of a head cannot become co-resident.  Forced here with
    the debug hook (every workgroup waits for a slab that does not exist): the give-up flags are set, the pre-pass outputs are
    NaN-poisoned, sageattn() returns NaN for every head with more than one slab -- never a plausible wrong number -- and the
    next call (hook off, fresh scratch) is sound again."""
    import sageattention_amd as sa
    lib = _cabi.load()
    q = torch.randn(1, 4, 1024, 128, device="cuda", dtype=torch.float16)
    k, v = _mk(1, 4, 1024, 128, torch.float16, "HND", 5)
    want = sa.sageattn(q, k, v, is_causal=False)
    torch.cuda.synchronize()
    sync = torch.empty(int(lib.sage_prepass_sync_words(1, 4)), dtype=torch.int32, device="cuda")
    lib.sage_debug_prepass_fail(1)
    try:
        got = quant.prepass_kv_fp8(k, v, "HND", smooth_k=True, sync=sync)
        torch.cuda.synchronize()
        assert quant.prepass_failed_heads(sync, 1, 4) == 8                  # K and V entry of each of the 4 heads
        assert torch.isnan(got[2]).all(), "k scales of a head that gave up must be NaN"
        assert torch.isnan(got[4]).all(), "v scales of a head that gave up must be NaN"
        assert (got[3].view(torch.uint8) == 0x7f).all(), "the V image of a head that gave up must be NaN bytes"
        quant._PrepassGuard.of(k.device).reset()      # (the launch above set the device's host word; this test wants the fused route again)
        o = sa.sageattn(q, k, v, is_causal=False)
        torch.cuda.synchronize()
        assert torch.isnan(o.float()).all(), "attention over a poisoned pre-pass must be NaN, not a number"
        old = quant._DEBUG
        quant._DEBUG = True
        try:
            with pytest.raises(_cabi.SageKernelError):
                quant.prepass_kv_fp8(k, v, "HND", smooth_k=True)
        finally:
            quant._DEBUG = old
        # a single-slab head has no barrier and is unaffected
        k1, v1 = _mk(1, 2, 300, 128, torch.float16, "HND", 6)
        r1 = quant.prepass_kv_fp8(k1, v1, "HND", smooth_k=True)
        assert torch.isfinite(r1[2]).all() and torch.isfinite(r1[4]).all()
    finally:
        lib.sage_debug_prepass_fail(0)
        quant._PrepassGuard.of(k.device).reset()            # (the guard's behaviour is the next test's subject)
    again = sa.sageattn(q, k, v, is_causal=False)
    torch.cuda.synchronize()
    assert torch.equal(again, want)


def test_a_give_up_reroutes_the_device_to_the_kernel_sequence():
    """Production behaviour (no SAGE_DEBUG): the workgroup that gives up also sets the caller's pinned host word; the next call on the
    device reads it (no synchronisation), warns once and takes the kernel sequence from then on -- its output is correct, and so is
    every later one, although the (forced) cause persists."""
    import warnings
    import sageattention_amd as sa
    lib = _cabi.load()
    q = torch.randn(1, 4, 1024, 128, device="cuda", dtype=torch.float16)
    k, v = _mk(1, 4, 1024, 128, torch.float16, "HND", 5)
    want = sa.sageattn(q, k, v, is_causal=True)
    qv = torch.randn(1500, 4, 128, device="cuda", dtype=torch.bfloat16)
    kv, vv = (torch.randn(1500, 2, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    cu = torch.tensor([0, 700, 1500], dtype=torch.int32, device="cuda")
    want_v = sa.sageattn_varlen(qv, kv, vv, cu, cu, 800, 800, is_causal=True)
    torch.cuda.synchronize()
    guard = quant._PrepassGuard.of(k.device)
    guard.reset()
    lib.sage_debug_prepass_fail(1)
    try:
        first = sa.sageattn(q, k, v, is_causal=True)                      # poisoned: its pre-pass gives up
        torch.cuda.synchronize()
        assert torch.isnan(first.float()).all()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            second = sa.sageattn(q, k, v, is_causal=True)
            third = sa.sageattn(q, k, v, is_causal=True)
            vl = sa.sageattn_varlen(qv, kv, vv, cu, cu, 800, 800, is_causal=True)
            torch.cuda.synchronize()
        assert len([x for x in w if "NaN-poisoned" in str(x.message)]) == 1, [str(x.message) for x in w]
        assert guard.tripped and not quant.prepass_fused_ok(k)
        assert torch.equal(second, want) and torch.equal(third, want) and torch.equal(vl, want_v)
    finally:
        lib.sage_debug_prepass_fail(0)
        guard.reset()
    assert quant.prepass_fused_ok(k)


def _cu_mask_stream(ncus_lo, ncus_hi):
    """A torch stream restricted to compute units [ncus_lo, ncus_hi) of XCD-interleaved numbering (hipExtStreamCreateWithCUMask)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in range(ncus_lo, ncus_hi):
        mask[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value), st, hip


def test_compute_units_held_by_another_stream_cost_one_call_not_the_process():
    """What the co-residency bound cannot see: sageattn() runs on a stream restricted to 8 compute units (so its heads may have 8 slabs)
    while a kernel on a second stream holds six of those eight for seconds.  The first call's pre-pass cannot get the slabs of a head
    resident together, gives up and returns NaN; from the second call on the device takes the kernel sequence: correct output, no
    SAGE_DEBUG, no hang."""
    import warnings
    import sageattention_amd as sa
    lib = _cabi.load()
    q = torch.randn(1, 2, 4096, 128, device="cuda", dtype=torch.bfloat16)
    k, v = _mk(1, 2, 4096, 128, torch.bfloat16, "HND", 31)            # 8 slabs per head
    want = sa.sageattn(q, k, v, is_causal=True)
    torch.cuda.synchronize()
    guard = quant._PrepassGuard.of(k.device)
    guard.reset()
    s8, h8, hip = _cu_mask_stream(0, 8)
    s6, h6, _ = _cu_mask_stream(0, 6)
    try:
        with torch.cuda.stream(s8):
            assert quant.prepass_fused_ok(k) and int(lib.sage_prepass_max_seqlen_stream(s8.cuda_stream)) == 8 * 512
            assert not quant.prepass_fused_ok(_mk(1, 1, 8192, 128, torch.bfloat16, "HND", 1)[0])     # 16 slabs: the sequence, by the stream-aware bound
        # two 1024-thread workgroups fill a compute unit's wave slots: 12 of them hold the six units for 2.5 s
        assert lib.sage_debug_spin(2500, 12, s6.cuda_stream) == 0
        with torch.cuda.stream(s8):
            first = sa.sageattn(q, k, v, is_causal=True)
        s8.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.cuda.stream(s8):
                second = sa.sageattn(q, k, v, is_causal=True)
                third = sa.sageattn(q, k, v, is_causal=True)
            torch.cuda.synchronize()
        if torch.isnan(first.float()).any():                   # the contention did bite (it does on an otherwise idle MI355X)
            assert guard.tripped and len([x for x in w if "NaN-poisoned" in str(x.message)]) == 1
        else:                                                  # the hardware found room after all: then nothing may have changed
            assert not guard.tripped and torch.equal(first, want)
        assert torch.equal(second, want) and torch.equal(third, want)
    finally:
        torch.cuda.synchronize()
        guard.reset()
        hip.hipStreamDestroy(h8)
        hip.hipStreamDestroy(h6)


@pytest.mark.parametrize("D,causal,layout,dtype,pv_accum,smooth_v,gqa", [
    (128, True, "HND", torch.bfloat16, "fp32", False, 1),
    (128, False, "NHD", torch.float16, "fp32", False, 4),
    (64, True, "HND", torch.float16, "fp32", False, 1),
    (64, False, "HND", torch.bfloat16, "fp16", True, 2),          # sub_mean + v_mean epilogue
    (128, True, "HND", torch.float16, "fp16", True, 1),
])
def test_fp16_pv_fused_q_is_bit_equal(D, causal, layout, dtype, pv_accum, smooth_v, gqa):
    """sageattn_qk_int8_pv_fp16_cuda quantises Q inside the attention kernel by default: same bits as the separate quantiser."""
    import warnings
    import sageattention_amd as sa
    g = torch.Generator(device="cuda").manual_seed(5)
    Hq, Hkv, Lq, Lk = 8, 8 // gqa, 1111, 1500
    mk = lambda h, l: (torch.randn((2, h, l, D) if layout == "HND" else (2, l, h, D), device="cuda", dtype=torch.float32, generator=g)).to(dtype)
    q, k, v = mk(Hq, Lq if not causal else Lk), mk(Hkv, Lk), mk(Hkv, Lk)
    kw = dict(tensor_layout=layout, is_causal=causal, qk_quant_gran="per_thread", pv_accum_dtype=pv_accum, smooth_v=smooth_v, return_lse=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        o0, l0 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, fuse_q_quant=False, **kw)
        o1, l1 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, **kw)
    _same(o1, o0, "o")
    _same(l1, l0, "lse")


def test_fp16_pv_kernel_is_repeatable_between_other_kernels():
    """Two identical FP16-PV calls must give identical bits whatever ran in between (guards the drain of the pipelined loop:
    its LDS-DMA once overwrote V fragments a slower wave was still reading, profiles/r2_run_r3l_fp16_drain_race.txt)."""
    import sageattention_amd as sa
    g = torch.Generator(device="cuda").manual_seed(9)
    q, k, v = (torch.randn(2, 32, 4096, 128, device="cuda", dtype=torch.float32, generator=g).to(torch.float16) for _ in range(3))
    v = torch.where(v.abs() < 2.0 ** -12, torch.full_like(v, 2.0 ** -12), v)
    x = torch.randn(4, 8, 777, 64, device="cuda", dtype=torch.bfloat16, generator=g)
    ref = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, smooth_k=False)
    for rep in range(12):
        if rep % 3 == 0:
            sa.sageattn(x, x, x, is_causal=bool(rep & 1))
        elif rep % 3 == 1:
            quant.prepass_kv_fp8(x, x, "HND", smooth_k=True, smooth_v=True)
        o = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, smooth_k=False)
        o2 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v * 2, is_causal=True, smooth_k=False)
        _same(o, ref, f"repeat {rep}")
        normal = ref.abs() >= 2.0 ** -13            # in the fp16 subnormal range round(2x) may differ from 2 round(x) by one quantum
        bad = (o2 != ref * 2) & normal
        assert not bool(bad.any()), f"V -> 2V, repeat {rep}: {int(bad.sum())} elements, first {bad.nonzero()[0].tolist()}"


@pytest.mark.parametrize("dt,D,layout,L,gran,blkk,smooth_v", [
    (0, 128, "HND", 1500, "per_thread", 64, False),
    (1, 64, "NHD", 700, "per_warp", 64, True),
    (1, 128, "HND", 2100, "per_thread", 128, True),
    (0, 64, "HND", 129, "per_warp", 128, False),
])
def test_fused_prepass_vs_oracle(oracle_mod, dt, D, layout, L, gran, blkk, smooth_v):
    """The one-launch kernel against the CPU oracle directly (oracle quantisers fed the kernel's own K mean / V mean, which are
    checked against the fp32 means): INT8 K, K scales, V scales and FP8 bytes bit-exact, padding tokens zero."""
    import numpy as np
    import util
    O = oracle_mod
    dtype = torch.float16 if dt == 0 else torch.bfloat16
    g = torch.Generator().manual_seed(31)
    B, H = 2, 3
    k = (torch.randn(B, H, L, D, generator=g) + 2.0 * torch.randn(1, H, 1, D, generator=g)).to(dtype)
    v = (torch.randn(B, H, L, D, generator=g) * (1 + 3 * torch.rand(1, H, 1, D, generator=g)) + 1.5).to(dtype)
    dev = lambda t: t.cuda() if layout == "HND" else t.cuda().transpose(1, 2).contiguous()
    hnd = lambda t: t if layout == "HND" else t.transpose(1, 2)
    km, k8, ks, vimg, vs, vm = quant.prepass_kv_fp8(dev(k), dev(v), layout, smooth_k=True, smooth_v=smooth_v, BLKK=blkk, qk_quant_gran=gran)
    torch.cuda.synchronize()
    # K mean: fp32 mean rounded once to the input dtype (half an ulp of slack for the summation order)
    want_km = k.double().mean(dim=2)
    ulp = 2.0 ** -10 if dt == 0 else 2.0 ** -7
    assert ((km.cpu().double() - want_km).abs() <= 0.51 * ulp * want_km.abs().clamp_min(2.0 ** -14) + 1e-7).all()
    style = O.STYLE_TRITON_THREAD if gran == "per_thread" else O.STYLE_CUDA
    gk, nk = O.group_index(L, gran, "k", blkk, blkk)
    rk8, rks = O.quant_int8(util.bits(k), dt, gk, nk, style=style, mean=util.bits(km.cpu()))
    assert (hnd(k8).cpu().numpy() == rk8).all() and (ks.cpu().numpy() == rks).all()
    mean = None
    if smooth_v:
        want_vm = O.v_mean_padded16(util.bits(v), dt)
        assert np.abs(vm.cpu().numpy() - want_vm).max() <= 1e-5 * max(1.0, float(np.abs(want_vm).max()))
        mean = vm.cpu().numpy()
    r8, rvs = O.quant_v_fp8(util.bits(v), dt, mean=mean)
    assert (vs.cpu().numpy() == rvs).all()
    assert (util.decode_v_image(vimg.cpu().numpy(), L, fp8=True) == r8).all()
    assert (util.decode_v_image(vimg.cpu().numpy(), vimg.shape[2] * 64, fp8=True)[..., L:, :] == 0).all()


@pytest.mark.parametrize("B,H,L,D,dtype,layout", [
    (2, 4, 1024, 128, torch.bfloat16, "HND"), (1, 3, 1000, 128, torch.float16, "NHD"), (2, 2, 77, 64, torch.bfloat16, "HND"),
    (1, 2, 2048 + 513, 64, torch.float16, "HND"), (1, 1, 4096, 128, torch.bfloat16, "NHD"),
])
def test_fused_prepass_fp16_image_bit_equals_prep_v_fp16(B, H, L, D, dtype, layout):
    k, v = _mk(B, H, L, D, dtype, layout, 3 * L + D)
    want_img = quant.prep_v_fp16(v, layout)
    ref = _sequence(k, v, layout, True, False, 64, "per_thread")
    for rep in range(2):
        km, k8, ks, vimg, vs, vm = quant.prepass_kv_fp8(k, v, layout, smooth_k=True, v_fp16=True)
        assert vs is None and vm is None
        _same(vimg, want_img, f"fp16 image (call {rep})")
        for a, b, name in zip((km, k8, ks), ref[:3], ("km", "k_int8", "k_scale")):
            _same(a, b, name)
