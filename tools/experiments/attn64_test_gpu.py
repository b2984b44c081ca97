"""GPU parity of the 256-row / one-wave-per-SIMD FP8 attention kernel (csrc/sage_attn64.hip), forced through the C ABI's route
switch (sage_set_attn64_mode): against the CPU oracle on identical operands (the bar of test_gpu_parity.py: max|diff| <=
2e-3 * max|o| + one output ulp, LSE <= 5e-3), against the 128-row kernel of the same entry point, and bit-repeatability over
back-to-back launches (the kernel's instruction order and hazards are hand-managed).

Shapes cover: ragged Lq / Lk (masked last tile, rows past Lq), causal diagonals in every wave position of a 256-row block,
causal with Lq != Lk (top-left aligned), GQA, every quantisation granularity of the INT8-Q entry point, the fused-Q entry point
in fp16 and bf16, smooth_v (v_mean epilogue), return_lse.
"""
import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import sageattention_amd as sa
    from sageattention_amd import _cabi, quant as sq
    DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _route():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    lib = _cabi.load()
    old = lib.sage_attn64_mode()
    lib.sage_set_attn64_mode(1)
    yield lib
    lib.sage_set_attn64_mode(old)


def T(dt):
    return torch.float16 if dt == 0 else torch.bfloat16


def rand_qkv(B, Hq, Hkv, Lq, Lk, dt, seed, kbias=1.0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Hq, Lq, 128, generator=g).to(T(dt))
    k = (torch.randn(B, Hkv, Lk, 128, generator=g) + kbias * torch.randn(1, Hkv, 1, 128, generator=g)).to(T(dt))
    v = torch.randn(B, Hkv, Lk, 128, generator=g).to(T(dt))
    return q, k, v


SHAPES = [  # B, Hq, Hkv, Lq, Lk, dt
    (1, 2, 2, 256, 256, 0), (2, 4, 2, 300, 300, 1), (1, 2, 1, 513, 777, 1), (1, 3, 3, 1024, 1024, 0),
    (1, 2, 2, 700, 64, 0), (1, 1, 1, 129, 1000, 1), (1, 2, 1, 1000, 129, 0), (1, 2, 2, 64, 64, 0), (1, 5, 5, 1, 1, 1),
    (1, 2, 2, 1281, 1500, 1),
]


@pytest.mark.parametrize("gran", ["per_thread", "per_warp", "per_block_fused"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"b{a}h{b}k{c}q{d}l{e}t{f}" for a, b, c, d, e, f in SHAPES])
def test_attn64_vs_oracle_and_the_128_row_kernel(oracle_mod, _route, shape, causal, gran):
    B, Hq, Hkv, Lq, Lk, dt = shape
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, dt, seed=Lq * 31 + Lk)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    fused = gran == "per_block_fused"           # third leg: the default route of sageattn() (fused Q quantisation, per-thread groups)
    g = "per_thread" if fused else gran

    def call():
        if fused:
            return sa.sageattn(qd, kd, vd, is_causal=causal, return_lse=True)
        return sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran=g, pv_accum_dtype="fp32+fp32",
                                               return_lse=True, fuse_q_quant=False)
    o1, lse1 = call()
    torch.cuda.synchronize()
    _route.sage_set_attn64_mode(0)
    o0, lse0 = call()
    torch.cuda.synchronize()
    _route.sage_set_attn64_mode(1)
    km = util.bits(sq.channel_mean(kd))
    ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8",
                                                qk_quant_gran=g, return_lse=True, km=km)
    got, old, ref = o1.float().cpu().numpy(), o0.float().cpu().numpy(), util.f32(ref, dt)
    assert np.isfinite(got).all()
    scale = float(np.abs(ref).max())
    ulp = (2 ** -8 if dt == 1 else 2 ** -11) * scale
    assert np.abs(got - ref).max() <= 2e-3 * scale + ulp, f"vs oracle: {np.abs(got - ref).max():.3e} (max|o| {scale:.3e})"
    assert np.abs(lse1.cpu().numpy() - lse_ref).max() <= 5e-3
    # the two kernels implement the same reference kernel: same operands, same P roundings; FP32 summation order differs
    assert np.abs(got - old).max() <= 2e-3 * scale + ulp
    assert np.abs((lse1 - lse0).cpu().numpy()).max() <= 1e-4


def test_attn64_smooth_v_epilogue(oracle_mod):
    q, k, v = rand_qkv(1, 4, 2, 600, 600, 0, seed=7)
    v = (v.float() + 2.0 * torch.randn(1, 2, 1, 128, generator=torch.Generator().manual_seed(8))).half()
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=True, qk_quant_gran="per_warp", pv_accum_dtype="fp32", smooth_v=True)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 0, is_causal=True, pv="f8",
                                          qk_quant_gran="per_warp", km=km, smooth_v=True)
    got, ref = o.float().cpu().numpy(), util.f32(ref, 0)
    scale = float(np.abs(ref).max())
    assert np.isfinite(got).all() and np.abs(got - ref).max() <= 2e-3 * scale + 2 ** -11 * scale


@pytest.mark.parametrize("causal", [False, True])
def test_attn64_is_bit_repeatable_under_load(causal):
    """200 back-to-back launches on two streams with another kernel competing for the CUs: every output equals the first one
    bit for bit (a missed hazard or an LDS race in the hand-scheduled loop shows up as a lane- or wave-sized difference)."""
    q, k, v = rand_qkv(2, 8, 4, 2048, 2048, 1, seed=11)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    first = sa.sageattn(qd, kd, vd, is_causal=causal)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    bad = 0
    for i in range(100):
        with torch.cuda.stream(side):
            (a @ a).sum()
            o2 = sa.sageattn(qd, kd, vd, is_causal=causal)
        o1 = sa.sageattn(qd, kd, vd, is_causal=causal)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o1, first)) + int(not torch.equal(o2, first))
    assert bad == 0, f"{bad} of 200 launches differ from the first"


def test_attn64_reads_no_stale_memory():
    """Every buffer the call allocates is pre-filled with NaN patterns by a poisoned allocator pass: the result must not change."""
    q, k, v = rand_qkv(1, 4, 4, 777, 777, 0, seed=5)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    want = sa.sageattn(qd, kd, vd, is_causal=True)
    torch.cuda.synchronize()
    del want_holder[:]
    want_holder.append(want.clone())
    del want
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 24,), float("nan"), device=DEV) for _ in range(8)]   # 512 MiB of NaN where the next allocations land
    del junk
    got = sa.sageattn(qd, kd, vd, is_causal=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want_holder[0])


want_holder = []
