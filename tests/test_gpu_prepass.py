"""The one-launch K / V pre-pass (sage_prepass_kv) against the kernel sequence it replaces: every output bit-equal.

The sequence (sage_channel_mean + sage_quant_qk_int8 + sage_prep_v_fp8) is itself pinned to the oracle by
tests/test_gpu_parity.py, so bit-equality here carries that parity over."""
import numpy as np
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
    sync = torch.zeros((int(_cabi.load().sage_prepass_sync_words(B, H)),), dtype=torch.int32, device="cuda")   # zeroed ONCE by its owner
    for rep in range(3):                   # the same buffer serves every call: the kernel returns its counters to zero (no zeroing launch)
        got = quant.prepass_kv_fp8(k, v, layout, smooth_k=smooth_k, smooth_v=smooth_v, BLKK=blkk, qk_quant_gran=gran, sync=sync)
        for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b, f"{name} (call {rep})")
    assert quant.prepass_failed_heads(sync, B, H) == 0
    assert int(sync.abs().sum().item()) == 0           # counters re-armed by the kernel, no give-up flag


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("SAGE_RANDOM_SEEDS", "100")))))
def test_random_shapes_fused_prepass_bit_equals_the_sequence(seed):
    """Seeded random shapes (lengths around the 16-token, 64-key and 512-token slab edges, up to a dozen slabs) of the comparison above."""
    rng = np.random.default_rng(7000 + seed)
    B, H = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    L = int(rng.choice([int(rng.integers(1, 130)), int(rng.integers(500, 530)), int(rng.integers(1000, 1100)), int(rng.integers(1, 6200))]))
    D = int(rng.choice([64, 128]))
    dtype = (torch.float16, torch.bfloat16)[int(rng.integers(0, 2))]
    layout = str(rng.choice(["HND", "NHD"]))
    smooth_k, smooth_v = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    gran = str(rng.choice(["per_thread", "per_warp", "per_block_triton"]))
    blkk = 64 if gran == "per_block_triton" else int(rng.choice([64, 128]))
    k, v = _mk(B, H, L, D, dtype, layout, seed)
    ref = _sequence(k, v, layout, smooth_k, smooth_v, blkk, gran)
    got = quant.prepass_kv_fp8(k, v, layout, smooth_k=smooth_k, smooth_v=smooth_v, BLKK=blkk, qk_quant_gran=gran)
    for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
        _same(a, b, f"{name}: B{B} H{H} L{L} D{D} {dtype} {layout} smooth_k{smooth_k} smooth_v{smooth_v} blkk{blkk} {gran}")


def test_the_per_stream_sync_buffers_are_bounded():
    """_stream_cache keeps one zeroed counter block per (purpose, device, stream); the table is bounded (streams come and go), locked, and a
    block is forgotten when the call it was handed to fails."""
    from sageattention_amd import _stream_cache as sc
    k, v = _mk(1, 2, 700, 64, torch.float16, "HND", 3)
    ref = quant.prepass_kv_fp8(k, v, "HND")
    with sc._LOCK:
        saved = dict(sc._CACHE)
    try:
        with sc._LOCK:
            for i in range(sc._MAX_ENTRIES):
                sc._CACHE[("prepass", 0, -1 - i)] = torch.zeros((4096,), dtype=torch.int32, device="cuda")
            full = len(sc._CACHE)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            got = quant.prepass_kv_fp8(k, v, "HND")
            key = sc._key("prepass", k.device)
            assert key in sc._CACHE and int(sc._CACHE[key].abs().max().item()) == 0       # back at zero behind the launch (the .item() synchronises)
            sc.drop("prepass", k.device)
            assert key not in sc._CACHE
        s.synchronize()
        assert full >= sc._MAX_ENTRIES and len(sc._CACHE) <= full        # the new stream's block displaced the oldest entry
        for a, b, name in zip(got, ref, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b, name)
    finally:
        with sc._LOCK:
            sc._CACHE.clear()
            sc._CACHE.update(saved)


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
    syncs = [(torch.zeros(words, dtype=torch.int32, device="cuda"), torch.zeros(words, dtype=torch.int32, device="cuda")) for _ in range(4)]
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


def test_a_workgroup_that_gives_up_computes_the_head_itself():
    """The in-launch head barrier stops waiting after a bounded time if the slabs of a head cannot become co-resident.  Forced here with
    the debug hook (every workgroup waits for a slab that does not exist): the give-up flags and the caller's host word are set, and
    EVERY output is still the same bits -- each workgroup re-read its head and summed the slabs in the same order -- for the three K
    conventions, both V images, smooth_v, D = 64 / 128, ragged lengths, and the packed (varlen) form."""
    import sageattention_amd as sa
    lib = _cabi.load()
    guard = quant._PrepassGuard.of(torch.device("cuda", 0))
    cases = [(1, 4, 1024, 128, torch.float16, "per_thread", False, False), (2, 3, 1300, 128, torch.bfloat16, "per_warp", True, False),
             (1, 2, 2100, 64, torch.bfloat16, "per_block_triton", False, True), (1, 5, 777, 64, torch.float16, "per_thread", True, False)]
    for B, H, L, D, dt, gran, smooth_v, v16 in cases:
        k, v = _mk(B, H, L, D, dt, "HND", 5 + L)
        want = quant.prepass_kv_fp8(k, v, "HND", smooth_k=True, smooth_v=smooth_v and not v16, qk_quant_gran=gran, v_fp16=v16)
        torch.cuda.synchronize()
        sync = torch.zeros(int(lib.sage_prepass_sync_words(B, H)), dtype=torch.int32, device="cuda")
        guard.reset()
        lib.sage_debug_prepass_fail(1)
        try:
            got = quant.prepass_kv_fp8(k, v, "HND", smooth_k=True, smooth_v=smooth_v and not v16, qk_quant_gran=gran, v_fp16=v16, sync=sync)
            torch.cuda.synchronize()
        finally:
            lib.sage_debug_prepass_fail(0)
        nflag = quant.prepass_failed_heads(sync, B, H)
        assert nflag == (B * H if v16 else 2 * B * H), nflag          # (the fp16 image has no statistics, hence no barrier)
        assert not guard.fused_allowed()                               # the host word was set by the kernel
        guard.reset()
        for a, b_, name in zip(got, want, ("km", "k_int8", "k_scale", "v_image", "v_scale", "v_mean")):
            _same(a, b_, f"{name} after a forced give-up ({B},{H},{L},{D},{gran})")
    # the packed form: K mean over all sequences' slabs
    lens = [700, 64, 1300, 5, 512]
    kp = (torch.randn(sum(lens), 3, 128, device="cuda") * 1.5 + torch.linspace(-2, 3, 128, device="cuda")).to(torch.bfloat16)
    vp = torch.randn(sum(lens), 3, 128, device="cuda").to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    plan = quant.varlen_plan(cu, cu, total_q=sum(lens), total_k=sum(lens), Hq=3, Hkv=3)
    want = quant.prepass_kv_varlen(kp, vp, cu, plan, max(lens))
    lib.sage_debug_prepass_fail(1)
    try:
        got = quant.prepass_kv_varlen(kp, vp, cu, plan, max(lens))
        torch.cuda.synchronize()
    finally:
        lib.sage_debug_prepass_fail(0)
        guard.reset()
    nblk = int(plan.cu_ks[-1].item())
    _same(got[0], want[0], "varlen km after a forced give-up")
    assert torch.equal(got[1], want[1]) and torch.equal(got[2][:nblk].view(torch.int32), want[2][:nblk].view(torch.int32))
    assert torch.equal(got[3][:nblk].view(torch.int16), want[3][:nblk].view(torch.int16))
    q = torch.randn(1, 4, 1024, 128, device="cuda", dtype=torch.float16)
    k, v = _mk(1, 4, 1024, 128, torch.float16, "HND", 5)
    want_o = sa.sageattn(q, k, v, is_causal=False)
    lib.sage_debug_prepass_fail(1)
    try:
        o = sa.sageattn(q, k, v, is_causal=False)
        torch.cuda.synchronize()
    finally:
        lib.sage_debug_prepass_fail(0)
        guard.reset()
    assert torch.equal(o, want_o), "attention over a pre-pass whose workgroups gave up must be the same bits"


def test_a_give_up_reroutes_the_device_to_the_kernel_sequence():
    """Production behaviour (no SAGE_DEBUG): the workgroup that gives up also sets the caller's pinned host word; the next call on the
    device reads it (no synchronisation), warns once and takes the kernel sequence from then on.  Every call is correct -- the one whose
    pre-pass gave up (its workgroups recomputed) and the later ones -- although the (forced) cause persists."""
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
        first = sa.sageattn(q, k, v, is_causal=True)                      # its pre-pass gives up and recomputes
        torch.cuda.synchronize()
        assert torch.equal(first, want)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            second = sa.sageattn(q, k, v, is_causal=True)
            third = sa.sageattn(q, k, v, is_causal=True)
            vl = sa.sageattn_varlen(qv, kv, vv, cu, cu, 800, 800, is_causal=True)
            torch.cuda.synchronize()
        assert len([x for x in w if "take the kernel sequence for the next" in str(x.message)]) == 1, [str(x.message) for x in w]
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


def test_compute_units_held_by_another_stream_cost_time_not_correctness():
    """What the co-residency bound cannot see: sageattn() runs on a stream restricted to 8 compute units (so its heads may have 8 slabs)
    while a kernel on a second stream holds six of those eight for seconds.  The first call's pre-pass cannot get the slabs of a head
    resident together: its workgroups stop waiting, recompute the head's statistics themselves, and the call returns the RIGHT bits;
    from the second call on the device takes the kernel sequence.  No SAGE_DEBUG, no hang, no NaN."""
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
        assert torch.equal(first, want), "a pre-pass that could not get its head resident must still be right"
        REPORT_TRIPPED.append(guard.tripped)
        if guard.tripped:                                      # the contention did bite (it does on an otherwise idle MI355X)
            assert len([x for x in w if "kernel sequence from now on" in str(x.message)]) == 1
        assert torch.equal(second, want) and torch.equal(third, want)
    finally:
        torch.cuda.synchronize()
        guard.reset()
        hip.hipStreamDestroy(h8)
        hip.hipStreamDestroy(h6)


REPORT_TRIPPED = []
