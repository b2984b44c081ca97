"""Soak of the hand-scheduled kernels (GPU): the steady-state loops of csrc/sage_attn_kernel.h pin their
instruction order in asm and manage their own hazards and LDS ring, so a missed hazard or a race shows up as a lane-, wave- or
tile-sized difference between two identical calls, rarely (round 2 shipped one that appeared once in a few hundred launches).

For every pipelined instantiation that a public entry point reaches -- FP8 PV D=128 / 64, causal / not, INT8-Q and fused-Q (fp16, bf16)
routes, FP16 PV D=128 / 64 in its CUDA and Triton forms, the split-KV route, the packed (varlen) route -- 200 launches, alternating between two
streams while a GEMM competes for the CUs, must equal the first launch bit for bit; and a call whose every temporary lands in
NaN-poisoned memory must equal a call on clean memory (nothing reads what it has not written).
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
_SOAK_ITERS = int(os.environ.get("SAGE_SOAK_ITERS", "100"))      # two launches per iteration (SAGE_SOAK_ITERS=1000: a one-off stress run)

if torch.cuda.is_available():
    import sageattention_amd as sa
    from sageattention_amd import _cabi
    DEV = torch.device("cuda:0")


def _qkv(B, Hq, Hkv, Lq, Lk, D, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Hq, Lq, D, generator=g).to(dtype).to(DEV)
    k = (torch.randn(B, Hkv, Lk, D, generator=g) + torch.randn(1, Hkv, 1, D, generator=g)).to(dtype).to(DEV)
    v = torch.randn(B, Hkv, Lk, D, generator=g).to(dtype).to(DEV)
    return q, k, v


F16, BF16 = torch.float16, torch.bfloat16
# name, entry point, kwargs, (B, Hq, Hkv, Lq, Lk, D, dtype), (unused)
CASES = [
    ("f8_d128_causal_fusedq_bf16", "sageattn", dict(is_causal=True), (2, 8, 4, 2048, 2048, 128, BF16), 0),
    ("f8_d128_noncausal_fusedq_f16", "sageattn", dict(is_causal=False), (1, 8, 8, 2048, 2048, 128, F16), 0),
    ("f8_d128_causal_int8q_per_thread", "fp8", dict(is_causal=True, qk_quant_gran="per_thread", pv_accum_dtype="fp32+fp32", fuse_q_quant=False),
     (1, 8, 8, 2048, 2048, 128, BF16), 0),
    ("f8_d128_noncausal_int8q_per_warp_single", "fp8", dict(is_causal=False, qk_quant_gran="per_warp", pv_accum_dtype="fp32"),
     (1, 8, 8, 1536, 1536, 128, F16), 0),
    ("f8_d64_causal_fusedq", "sageattn", dict(is_causal=True), (2, 8, 8, 2048, 2048, 64, BF16), 0),
    ("f8_d64_noncausal_int8q", "fp8", dict(is_causal=False, qk_quant_gran="per_thread", pv_accum_dtype="fp32+fp32", fuse_q_quant=False),
     (1, 8, 8, 2048, 2048, 64, F16), 0),
    ("f16_d128_causal_fusedq", "fp16", dict(is_causal=True, qk_quant_gran="per_thread", pv_accum_dtype="fp32"), (1, 8, 8, 2048, 2048, 128, F16), 0),
    ("f16_d128_noncausal_per_warp_cuda_form", "fp16", dict(is_causal=False, qk_quant_gran="per_warp", pv_accum_dtype="fp16+fp32"),
     (1, 8, 4, 2048, 2048, 128, BF16), 0),
    ("f16_d64_causal_int8q", "fp16", dict(is_causal=True, qk_quant_gran="per_thread", pv_accum_dtype="fp32", fuse_q_quant=False),
     (1, 8, 8, 2048, 2048, 64, F16), 0),
    ("f16_d128_triton_form_causal", "triton", dict(is_causal=True), (1, 8, 8, 2048, 2048, 128, F16), 0),
    ("f16_d64_triton_form_noncausal", "triton", dict(is_causal=False), (1, 8, 8, 2048, 2048, 64, BF16), 0),
    ("f8_split_kv_cross_attention", "fp8", dict(is_causal=False, qk_quant_gran="per_thread", pv_accum_dtype="fp32+fp32", split_kv=4),
     (1, 8, 8, 128, 4096, 128, BF16), 0),
]


def _fn(entry):
    return {"sageattn": sa.sageattn, "fp8": sa.sageattn_qk_int8_pv_fp8_cuda, "fp16": sa.sageattn_qk_int8_pv_fp16_cuda,
            "triton": sa.sageattn_qk_int8_pv_fp16_triton}[entry]


@pytest.fixture
def route():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _cabi.load()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_200_launches_on_two_streams_are_bit_identical(route, case):
    name, entry, kw, shape, mode = case
    q, k, v = _qkv(*shape, seed=len(name))
    fn = _fn(entry)
    first = fn(q, k, v, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(first.float()).all()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    bad = 0
    for i in range(_SOAK_ITERS):
        with torch.cuda.stream(side):
            if i % 4 == 0:
                (a @ a).sum()                      # a competing kernel on the other stream
            o2 = fn(q, k, v, **kw)
        o1 = fn(q, k, v, **kw)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o1, first)) + int(not torch.equal(o2, first))
    assert bad == 0, f"{name}: {bad} of {2 * _SOAK_ITERS} launches differ from the first"


@pytest.mark.parametrize("case", [CASES[i] for i in (0, 4, 6, 9, 11)], ids=[CASES[i][0] for i in (0, 4, 6, 9, 11)])
def test_results_do_not_depend_on_what_the_allocator_hands_out(route, case):
    """Every temporary of the call (quantised operands, scales, V image, workspaces, partial outputs) is allocated with torch.empty.
    Once with the caching allocator's free blocks full of NaN patterns, once full of 0x5A bytes: same bits as a call on fresh memory."""
    name, entry, kw, shape, mode = case
    q, k, v = _qkv(*shape, seed=3 + len(name))
    fn = _fn(entry)
    want = fn(q, k, v, **kw).clone()
    torch.cuda.synchronize()
    for fill in (float("nan"), None):
        torch.cuda.empty_cache()
        junk = [torch.empty(1 << 24, device=DEV) for _ in range(16)]          # 1 GiB of blocks the next allocations are cut from
        for j in junk:
            if fill is None:
                j.view(torch.uint8).fill_(0x5A)
            else:
                j.fill_(fill)
        del junk
        got = fn(q, k, v, **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"{name}: the result depends on stale memory ({'NaN' if fill is not None else '0x5A'} fill)"


@pytest.mark.parametrize("shape", [
    (2, 32, 32, 1024, 128),     # 64 items per XCD: one round, folded (a long and a short block per CU)
    (1, 8, 8, 6144, 128),       # 48 items per XCD, one head: folded, partial second half
    (1, 8, 2, 3968, 128),       # 31 items per XCD: one workgroup per CU, no fold
    (2, 8, 8, 2048, 128),       # two heads per XCD, several rounds
    (1, 24, 24, 1500, 64),      # D=64 (three workgroups per CU), three heads per XCD, ragged last block
    (1, 16, 16, 777, 128),      # ragged
    (1, 12, 12, 2048, 128),     # 12 heads: one whole head per XCD + four left-over heads dealt to all XCDs by query block
    (1, 4, 4, 16384, 128),      # fewer heads than XCDs: every head dealt by query block, 64 items per XCD, folded
    (1, 28, 4, 1100, 128),      # GQA, 3 whole + 4 left-over heads, 9 blocks (ragged octet: empty workgroups exit)
    (1, 7, 7, 1300, 64),        # D=64, left-over heads only
], ids=lambda s: "b%dh%dk%dn%dd%d" % s)
def test_causal_work_order_does_not_change_a_bit(route, shape):
    """The causal launches pick which (head, query block) a workgroup takes from blockIdx (grouped / folded by grid size,
    sage_set_work_order); every order must be a permutation of the same work: outputs and LSE bit-equal to the head-major order."""
    B, Hq, Hkv, N, D = shape
    q, k, v = _qkv(B, Hq, Hkv, N, N, D, BF16, seed=N)
    old = route.sage_work_order()
    try:
        outs = {}
        for order in (0, -1, 1, 3, 5, 64):
            route.sage_set_work_order(order)
            o, lse = sa.sageattn(q, k, v, is_causal=True, return_lse=True)
            o2 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, qk_quant_gran="per_warp", pv_accum_dtype="fp32")
            torch.cuda.synchronize()
            outs[order] = (o, lse, o2)
        assert torch.isfinite(outs[0][0].float()).all()
        for order, (o, lse, o2) in outs.items():
            assert torch.equal(o, outs[0][0]) and torch.equal(lse, outs[0][1]) and torch.equal(o2, outs[0][2]), f"order {order} differs"
    finally:
        route.sage_set_work_order(old)


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_200_launches_on_two_streams_are_bit_identical(route, causal):
    """sageattn_varlen's default route: index arrays from sage_varlen_plan, per-block Q quantised in the attention prologue (a workgroup-wide
    abs-max through LDS before the first tile), K / V halves as separate kernels -- every launch equals the first one bit for bit."""
    lens = [2048, 1, 640, 129, 3000, 64, 1500, 777]
    total = sum(lens)
    g = torch.Generator().manual_seed(17)
    q = torch.randn(total, 8, 128, generator=g).to(BF16).to(DEV)
    k = (torch.randn(total, 4, 128, generator=g) + torch.randn(1, 4, 128, generator=g)).to(BF16).to(DEV)
    v = torch.randn(total, 4, 128, generator=g).to(BF16).to(DEV)
    cu = torch.nn.functional.pad(torch.tensor(lens).cumsum(0), (1, 0)).to(torch.int32).to(DEV)
    fn = lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
    first = fn()
    torch.cuda.synchronize()
    assert torch.isfinite(first.float()).all()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    bad = 0
    for i in range(100):
        with torch.cuda.stream(side):
            if i % 4 == 0:
                (a @ a).sum()
            o2 = fn()
        o1 = fn()
        torch.cuda.synchronize()
        bad += int(not torch.equal(o1, first)) + int(not torch.equal(o2, first))
    assert bad == 0, f"{bad} of 200 varlen launches differ from the first"


# ------------------------------------------------------------------------------------------------ persistent launches (ticket queues)
# name, entry point, kwargs, (B, Hq, Hkv, Lq, Lk, D, dtype): non-causal calls of at least two rounds of workgroups (D = 128: 2 x 512 items,
# D = 64: 2 x 768), which ops.launch_hooks(force_persistent=True) sends down the ticket route (SAGE_ATTR_FORCE_PERSISTENT; the default threshold is twelve
# rounds, i.e. shapes five times larger -- the same kernel, loop and queues)
PERSISTENT_CASES = [
    ("f8_d128_fusedq_bf16", "sageattn", dict(), (1, 8, 8, 16384, 16384, 128, BF16)),
    ("f8_d64_fusedq_f16", "sageattn", dict(), (1, 12, 12, 16384, 16384, 64, F16)),
    ("f8_d64_int8q_per_thread", "fp8", dict(qk_quant_gran="per_thread", pv_accum_dtype="fp32+fp32", fuse_q_quant=False), (1, 12, 6, 16384, 16384, 64, BF16)),
    ("f8_d128_int8q_per_warp_single_exact", "fp8", dict(qk_quant_gran="per_warp", pv_accum_dtype="fp32", fp8_scores="exact"), (1, 8, 8, 16384, 16384, 128, F16)),
    ("f16_d128_triton_form", "triton", dict(), (1, 8, 8, 16384, 16384, 128, F16)),
    ("f16_d64_cuda_form_gqa", "fp16", dict(qk_quant_gran="per_thread", pv_accum_dtype="fp32"), (1, 12, 4, 16384, 16384, 64, BF16)),
]


@pytest.fixture
def forced_persistent(monkeypatch):
    import ctypes
    from sageattention_amd import ops
    probe = ctypes.c_int32(-1)
    monkeypatch.setattr(ops, "_PERSISTENT", True)
    with ops.launch_hooks(grid_probe=probe, force_persistent=True):       # (gone when the test ends, also on a failure)
        yield ops, probe


def _soak(fn, want, n=None):
    n = _SOAK_ITERS if n is None else n
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    bad = 0
    for i in range(n):
        with torch.cuda.stream(side):
            if i % 4 == 0:
                (a @ a).sum()                      # a competing kernel on the other stream
            o2 = fn()
        o1 = fn()
        torch.cuda.synchronize()
        bad += int(not torch.equal(o1, want)) + int(not torch.equal(o2, want))
    return bad


@pytest.mark.parametrize("case", PERSISTENT_CASES, ids=[c[0] for c in PERSISTENT_CASES])
def test_persistent_route_200_launches_on_two_streams_equal_the_ordinary_launch(route, forced_persistent, monkeypatch, case):
    """The ticket-queue route of the large non-causal launches: 200 launches alternating between two streams beside a
    competing GEMM, each stream with ONE self-cleaning counter block for all its launches (round 6), each bit-identical to the ORDINARY launch of the same call; the probe confirms
    that the route is really taken (fewer workgroups than work items) and really off for the reference run."""
    ops, probe = forced_persistent
    name, entry, kw, shape = case
    q, k, v = _qkv(*shape, seed=11 + len(name))
    fn = lambda: _fn(entry)(q, k, v, is_causal=False, **kw)
    items = shape[0] * shape[1] * ((shape[3] + 127) // 128)
    monkeypatch.setattr(ops, "_PERSISTENT", False)
    want = fn()
    torch.cuda.synchronize()
    assert probe.value == items and torch.isfinite(want.float()).all()
    monkeypatch.setattr(ops, "_PERSISTENT", True)
    got = fn()
    torch.cuda.synchronize()
    assert 0 < probe.value < items and probe.value % 32 == 0, (probe.value, items)
    assert torch.equal(got, want)
    bad = _soak(fn, want)
    assert bad == 0, f"{name}: {bad} of 200 persistent launches differ from the ordinary launch"


@pytest.mark.parametrize("causal", [False, True])
def test_persistent_route_varlen_200_launches_equal_the_ordinary_launch(route, forced_persistent, monkeypatch, causal):
    """The packed (varlen) launch over the device-built work list on the ticket route: the logical grid comes from the plan's header on the
    device, items differ in length by two orders of magnitude.  Causal too (round 5): the causal kernels of this route carry the loop."""
    ops, probe = forced_persistent
    lens = [4096, 1, 640, 129, 3000, 64, 1500, 777, 2100]
    total = sum(lens)
    g = torch.Generator().manual_seed(23)
    q = torch.randn(total, 16, 128, generator=g).to(BF16).to(DEV)
    k = (torch.randn(total, 4, 128, generator=g) + torch.randn(1, 4, 128, generator=g)).to(BF16).to(DEV)
    v = torch.randn(total, 4, 128, generator=g).to(BF16).to(DEV)
    cu = torch.nn.functional.pad(torch.tensor(lens).cumsum(0), (1, 0)).to(torch.int32).to(DEV)
    fn = lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
    monkeypatch.setattr(ops, "_PERSISTENT", False)
    want = fn()
    torch.cuda.synchronize()
    ordinary = probe.value
    monkeypatch.setattr(ops, "_PERSISTENT", True)
    got = fn()
    torch.cuda.synchronize()
    assert 0 < probe.value < ordinary and probe.value % 32 == 0, (probe.value, ordinary)
    assert torch.equal(got, want)
    bad = _soak(fn, want)
    assert bad == 0, f"{bad} of 200 persistent varlen launches differ from the ordinary launch"


def test_persistent_route_in_a_captured_graph_replays_identically(route, forced_persistent, monkeypatch):
    """A persistent launch captured in a HIP graph: the workspace is allocated and zeroed inside the capture (a fill node in front of the
    attention node), so every replay starts from zero counters -- 20 replays equal the eager ordinary launch, also after the inputs changed."""
    ops, probe = forced_persistent
    q, k, v = _qkv(1, 12, 12, 16384, 16384, 64, BF16, seed=5)
    sa.sageattn(q, k, v)                                  # warm the allocator outside the capture
    torch.cuda.synchronize()
    assert 0 < probe.value < 12 * 128
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = sa.sageattn(q, k, v)
    monkeypatch.setattr(ops, "_PERSISTENT", False)
    for rep in range(2):
        want = sa.sageattn(q, k, v)
        torch.cuda.synchronize()
        assert probe.value == 12 * 128
        for _ in range(10):
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want), f"replay differs (inputs version {rep})"
        q.copy_(torch.roll(q, 7, dims=2)); k.mul_(0.5); v.add_(0.25)          # new inputs in place: the graph reads the same addresses
