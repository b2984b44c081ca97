"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the reference's golden
vectors and fp32 SDPA.  Run on an MI355X:  python -m pytest tests -m gpu -q

Bars (stated here, used below):
  * INT8 tensors, scales, FP8 bytes, fp16 V image: BIT-EXACT vs the oracle / the reference fixtures.
  * attention output vs the oracle on identical quantised operands AND THE SAME SCHEDULE: max|diff| <= 2e-3 * max|o|
    (differences: v_exp_f32 vs exp2f, FP32 summation order, FP32 instead of FP16 tile products).  "Same schedule": the oracle mode
    that mirrors the route rounding for rounding -- FP8 score form folded / exact (oracle score_mode), split-KV (_split_oracle).
  * an FP8 schedule variant that is a DEFAULT route (the folded score form): additionally rel-RMS <= 1e-2 against the exact form and
    cos / rel-RMSE vs fp32 SDPA within 1e-4 / 1e-3 of the exact form's (test_fp8_score_forms_*).
  * vs the reference Triton outputs (golden fixtures): max|diff| <= 2e-3 * max|o| (fp16),
    one bf16 ulp more for bf16 outputs (the CPU interpreter truncates fp32->bf16).
  * vs fp32 SDPA on randn inputs: cos >= 0.9995 / rel-RMSE <= 2% (FP16 PV), cos >= 0.999 / <= 5% (FP8 PV).
"""
import json
import os

import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import sageattention_amd as sa
    from sageattention_amd import _cabi, ops as sa_ops, quant as sq
    DEV = torch.device("cuda:0")
    # the FP8 score form the product runs by default: "exact", the reference's formula (SAGE_FP8_SCORES=folded runs the whole suite on the
    # opt-in variant against the oracle mode that mirrors it).  Every default-route comparison below is against the EXACT oracle.
    SCORES = "folded" if sa_ops._FP8_FOLDED else "exact"
else:
    SCORES = "exact"

REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _gpu_and_report():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _cabi.load()   # fail loudly if the HIP extension is missing
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def oracle_km_packed(k_bits, dt):
    """k.mean(dim=0, keepdim=True) of a packed [T,H,D] tensor in the input dtype (core.py:433), on the host."""
    import oracle
    kf = oracle.to_f32(k_bits, dt)
    return oracle.convert(kf.astype(np.float64).mean(axis=0, keepdims=True).astype(np.float32), "f16" if dt == 0 else "bf16")


def T(dtype_code):
    return torch.float16 if dtype_code == 0 else torch.bfloat16


def rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed, kbias=0.0, layout="HND"):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Hq, Lq, D, generator=g).to(T(dt))
    k = (torch.randn(B, Hkv, Lk, D, generator=g) + kbias * torch.randn(1, Hkv, 1, D, generator=g)).to(T(dt))
    v = torch.randn(B, Hkv, Lk, D, generator=g).to(T(dt))
    return q, k, v


def to_dev(t, layout):
    """HND host tensor -> device tensor in the requested layout (NHD = physically transposed)."""
    t = t.to(DEV)
    return t if layout == "HND" else t.transpose(1, 2).contiguous()


def to_hnd(t, layout):
    return t if layout == "HND" else t.transpose(1, 2)


# ------------------------------------------------------------------------------------------------ quant
@pytest.mark.parametrize("gran", ["per_block_triton", "per_block_cuda", "per_warp32", "per_warp16", "per_thread"])
@pytest.mark.parametrize("dt,D,layout", [(0, 128, "HND"), (1, 64, "NHD"), (0, 64, "HND"), (1, 128, "NHD")])
def test_quant_int8_bit_exact(oracle_mod, gran, dt, D, layout):
    _check_quant_int8(oracle_mod, gran, dt, D, layout, 2, 4, 2, 300, 333, seed=1, kbias=2.0)


def _check_quant_int8(oracle_mod, gran, dt, D, layout, B, Hq, Hkv, Lq, Lk, seed, kbias):
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=seed, kbias=kbias)
    km = k.float().mean(dim=2).to(T(dt))                                   # [B,Hkv,D]
    qd, kd = to_dev(q, layout), to_dev(k, layout)
    kmd = km.to(DEV).unsqueeze(2 if layout == "HND" else 1)
    O = oracle_mod
    sm = D ** -0.5
    if gran.startswith("per_block"):
        backend = gran.split("_")[-1]
        q8, qs, k8, ks = sq.per_block_int8(qd, kd, km=kmd, sm_scale=sm, tensor_layout=layout, quantization_backend=backend)
        style = O.STYLE_TRITON if backend == "triton" else O.STYLE_CUDA
        gq, nq = O.group_index(Lq, "per_block", "q", 128, 128); gk, nk = O.group_index(Lk, "per_block", "k", 64, 64)
        rq8, rqs = O.quant_int8(util.bits(q), dt, gq, nq, pre_scale=np.float32(sm * O.LOG2E), style=style)
        rk8, rks = O.quant_int8(util.bits(k), dt, gk, nk, style=style, mean=util.bits(km))
    elif gran.startswith("per_warp"):
        W = int(gran[-2:])
        q8, qs, k8, ks = sq.per_warp_int8(qd, kd, kmd, WARPQ=W, tensor_layout=layout)
        gq, nq = O.group_index(Lq, "per_warp", "q", 128, W); gk, nk = O.group_index(Lk, "per_warp", "k", 64, 64)
        rq8, rqs = O.quant_int8(util.bits(q), dt, gq, nq, style=O.STYLE_CUDA)
        rk8, rks = O.quant_int8(util.bits(k), dt, gk, nk, style=O.STYLE_CUDA, mean=util.bits(km))
    else:
        q8, qs, k8, ks = sq.per_thread_int8(qd, kd, kmd, tensor_layout=layout)
        gq, nq = O.group_index(Lq, "per_thread", "q", 128, 32); gk, nk = O.group_index(Lk, "per_thread", "k", 64, 64)
        rq8, rqs = O.quant_int8(util.bits(q), dt, gq, nq, style=O.STYLE_TRITON_THREAD)
        rk8, rks = O.quant_int8(util.bits(k), dt, gk, nk, style=O.STYLE_TRITON_THREAD, mean=util.bits(km))
    torch.cuda.synchronize()
    assert q8.shape == qd.shape and k8.shape == kd.shape and q8.dtype == torch.int8
    got = [to_hnd(q8, layout).cpu().numpy(), qs.cpu().numpy(), to_hnd(k8, layout).cpu().numpy(), ks.cpu().numpy()]
    for name, a, b in zip(("q_int8", "q_scale", "k_int8", "k_scale"), got, (rq8, rqs, rk8, rks)):
        assert a.shape == b.shape, name
        assert (a == b).all(), f"{name}: {(a != b).sum()} mismatches of {a.size}"


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SAGE_RANDOM_SEEDS", "100")))))
def test_random_operands_bit_exact_vs_oracle(oracle_mod, seed):
    """Seeded random shapes through the operand kernels: INT8 Q / K and their scales (every granularity and rounding style), the FP8 V image
    and its per-channel scales, the FP16 V image -- every byte against the oracle."""
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.choice([64, 128]))
    Hkv = int(rng.integers(1, 4))
    Hq = Hkv * int(rng.choice([1, 2, 4]))
    B = int(rng.integers(1, 3))
    pick_len = lambda: int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(190, 330)), int(rng.integers(500, 1500))]))
    Lq, Lk = pick_len(), pick_len()
    dt = int(rng.integers(0, 2))
    layout = str(rng.choice(["HND", "NHD"]))
    gran = str(rng.choice(["per_block_triton", "per_block_cuda", "per_warp32", "per_warp16", "per_thread"]))
    _check_quant_int8(oracle_mod, gran, dt, D, layout, B, Hq, Hkv, Lq, Lk, seed=seed, kbias=float(3 * rng.random()))
    _check_v_images(oracle_mod, dt, D, layout, Lk, B, Hkv, seed=seed)


@pytest.mark.parametrize("style", ["triton", "thread"])
def test_quant_division_is_correctly_rounded_stress(oracle_mod, style):
    """The quantiser replaces x/scale by a reciprocal + two FMA refinements; it must stay bit-exact with true IEEE
    division, including values parked right on the +-0.5 rounding boundaries (x = (k + 0.5) * scale +- 1 ulp)."""
    g = torch.Generator().manual_seed(99)
    B, H, L, D = 2, 8, 2048, 128
    x = torch.randn(B, H, L, D, generator=g) * (1.0 + 10.0 * torch.rand(B, H, 1, 1, generator=g))
    # adversarial rows: multiples of amax/127 offset by half a step, nudged by one fp16 ulp either way
    amax = x.abs().amax(dim=(2, 3), keepdim=True)
    kk = torch.randint(-127, 127, (B, H, L // 4, D), generator=g).float()
    x[:, :, : L // 4] = (kk + 0.5) * (amax / 127.0)
    x = x.half()
    x[:, :, : L // 8] = torch.nextafter(x[:, :, : L // 8], torch.full_like(x[:, :, : L // 8], 1e4))
    xd = x.to(DEV)
    if style == "triton":
        q8, qs, k8, ks = sq.per_block_int8(xd, xd, sm_scale=1.0 / sq.LOG2E)
        gq, nq = oracle_mod.group_index(L, "per_block", "q", 128, 128); gk, nk = oracle_mod.group_index(L, "per_block", "k", 64, 64)
        rq8, rqs = oracle_mod.quant_int8(util.bits(x), 0, gq, nq, pre_scale=np.float32((1.0 / sq.LOG2E) * oracle_mod.LOG2E), style=oracle_mod.STYLE_TRITON)
        rk8, rks = oracle_mod.quant_int8(util.bits(x), 0, gk, nk, style=oracle_mod.STYLE_TRITON)
    else:
        q8, qs, k8, ks = sq.per_thread_int8(xd, xd)
        gq, nq = oracle_mod.group_index(L, "per_thread", "q", 128, 32); gk, nk = oracle_mod.group_index(L, "per_thread", "k", 64, 64)
        rq8, rqs = oracle_mod.quant_int8(util.bits(x), 0, gq, nq, style=oracle_mod.STYLE_TRITON_THREAD)
        rk8, rks = oracle_mod.quant_int8(util.bits(x), 0, gk, nk, style=oracle_mod.STYLE_TRITON_THREAD)
    torch.cuda.synchronize()
    assert (qs.cpu().numpy() == rqs).all() and (ks.cpu().numpy() == rks).all()
    assert (q8.cpu().numpy() == rq8).all(), f"{(q8.cpu().numpy() != rq8).sum()} q mismatches"
    assert (k8.cpu().numpy() == rk8).all(), f"{(k8.cpu().numpy() != rk8).sum()} k mismatches"


def test_quant_golden_per_thread(oracle_mod):
    """Bit-exact against the reference's own per-thread quantiser output (fixture)."""
    z, (B, Hq, Hkv, Lq, Lk, D, dt, _) = util.golden("per_thread_quant_d128_f16")
    q, k, km = (util.from_bits(z[n], dt, DEV) for n in ("q", "k", "km"))
    q8, qs, k8, ks = sq.per_thread_int8(q, k, km)
    for a, b in ((q8, "q_int8"), (qs, "q_scale"), (k8, "k_int8"), (ks, "k_scale")):
        assert (a.cpu().numpy() == z[b]).all(), b


@pytest.mark.parametrize("name", ["per_thread_sm90_d128_f16", "per_thread_sm90_d64_bf16", "per_thread_warpq16_d128_f16"])
def test_quant_golden_per_thread_groups(name):
    """Bit-exact against the reference's per-thread quantiser in the sm90 groups (q per 16 of 64 rows, k per 128 keys: core.py:967) and with
    WARPQ = 16 in 128-row blocks (core.py:604) -- reference outputs, tests/golden/gen_golden.py."""
    z, (B, Hq, Hkv, Lq, Lk, D, dt, _) = util.golden(name)
    BLKQ, WARPQ, BLKK, WARPK = (int(x) for x in z["groups"])
    q, k, km = (util.from_bits(z[n], dt, DEV) for n in ("q", "k", "km"))
    q8, qs, k8, ks = sq.per_thread_int8(q, k, km, BLKQ=BLKQ, WARPQ=WARPQ, BLKK=BLKK, WARPK=WARPK)
    for a, b in ((q8, "q_int8"), (qs, "q_scale"), (k8, "k_int8"), (ks, "k_scale")):
        assert a.shape == z[b].shape and (a.cpu().numpy() == z[b]).all(), b


@pytest.mark.parametrize("dt,D,layout,L", [(0, 128, "HND", 300), (1, 64, "NHD", 64), (1, 128, "HND", 1000), (0, 64, "NHD", 129)])
def test_prep_v_images_bit_exact(oracle_mod, dt, D, layout, L):
    _check_v_images(oracle_mod, dt, D, layout, L, 2, 3, seed=5)


def _check_v_images(oracle_mod, dt, D, layout, L, B, H, seed):
    g = torch.Generator().manual_seed(seed)
    v = (torch.randn(B, H, L, D, generator=g) * (1 + 3 * torch.rand(1, H, 1, D, generator=g))).to(T(dt))
    vd = to_dev(v, layout)
    img8, vs, _ = sq.per_channel_fp8(vd, tensor_layout=layout)
    img16 = sq.prep_v_fp16(vd, tensor_layout=layout)
    torch.cuda.synchronize()
    r8, rvs = oracle_mod.quant_v_fp8(util.bits(v), dt)
    assert (vs.cpu().numpy() == rvs).all()
    got8 = util.decode_v_image(img8.cpu().numpy(), L, fp8=True)
    assert (got8 == r8).all(), f"{(got8 != r8).sum()} fp8 mismatches"
    # padding tokens are zero
    full = util.decode_v_image(img8.cpu().numpy(), img8.shape[2] * 64, fp8=True)
    assert (full[..., L:, :] == 0).all()
    r16 = util.bits(v) if dt == 0 else oracle_mod.convert(util.f32(util.bits(v), dt), "f16")
    got16 = util.decode_v_image(img16.cpu().view(torch.int16).numpy().view(np.uint16), L, fp8=False)
    assert (got16 == r16).all()


@pytest.mark.parametrize("dt,D,layout,L", [(0, 128, "HND", 1500), (1, 64, "NHD", 513), (1, 128, "NHD", 64), (0, 64, "HND", 7)])
def test_channel_mean_matches_torch_and_is_deterministic(dt, D, layout, L):
    """K-smoothing mean as a HIP reduction (replaces torch's k.mean, core.py:280): fp32 accumulate, one rounding."""
    g = torch.Generator().manual_seed(8)
    k = (torch.randn(2, 3, L, D, generator=g) + 2.0 * torch.randn(1, 3, 1, D, generator=g)).to(T(dt))
    kd = to_dev(k, layout)
    a = sq.channel_mean(kd, layout)
    b = sq.channel_mean(kd, layout)
    torch.cuda.synchronize()
    assert a.shape == (2, 3, D) and a.dtype == k.dtype and torch.equal(a, b)
    want = k.double().mean(dim=2)
    ulp = (2.0 ** -10 if dt == 0 else 2.0 ** -7) * want.abs().clamp_min(2.0 ** -14)
    assert ((a.cpu().double() - want).abs() <= 0.51 * ulp + 1e-7).all()      # correctly rounded up to fp32 summation error
    # packed [sum L, H, D] form used by sageattn_varlen
    kp = kd if layout == "NHD" else kd.transpose(1, 2).contiguous()
    m = sq.channel_mean_packed(kp.reshape(-1, 3, D))
    want_p = k.double().mean(dim=(0, 2))
    assert m.shape == (1, 3, D) and ((m[0].cpu().double() - want_p).abs() <= 0.51 * (2.0 ** -10 if dt == 0 else 2.0 ** -7) * want_p.abs().clamp_min(2.0 ** -14) + 1e-7).all()


@pytest.mark.parametrize("dt,D,layout,L", [(0, 128, "HND", 300), (1, 64, "NHD", 129)])
def test_prep_v_fp8_smooth_v_bit_exact(oracle_mod, dt, D, layout, L):
    g = torch.Generator().manual_seed(6)
    v = (torch.randn(2, 2, L, D, generator=g) + 4.0 * torch.randn(1, 2, 1, D, generator=g)).to(T(dt))
    img8, vs, vm = sq.per_channel_fp8(to_dev(v, layout), tensor_layout=layout, smooth_v=True)
    torch.cuda.synchronize()
    vm_h = vm.cpu().numpy()
    want_vm = oracle_mod.v_mean_padded16(util.bits(v), dt)                    # sum / ceil16(L), fused.cu:335,381
    assert np.abs(vm_h - want_vm).max() <= 1e-5 * max(1.0, float(np.abs(want_vm).max()))
    r8, rvs = oracle_mod.quant_v_fp8(util.bits(v), dt, mean=vm_h)            # same mean -> bit-exact bytes
    assert (vs.cpu().numpy() == rvs).all()
    got8 = util.decode_v_image(img8.cpu().numpy(), L, fp8=True)
    assert (got8 == r8).all(), f"{(got8 != r8).sum()} fp8 mismatches"
    full = util.decode_v_image(img8.cpu().numpy(), img8.shape[2] * 64, fp8=True)
    assert (full[..., L:, :] == 0).all()                                       # padding stays zero, not -mean


@pytest.mark.parametrize("pv", ["f8", "f16"])
@pytest.mark.parametrize("causal", [False, True])
def test_smooth_v_paths_vs_oracle_and_sdpa(oracle_mod, pv, causal):
    """smooth_v: reference core.py:617-619 (fp16 accumulate API) and :811-813 (fp8, pv_accum_dtype="fp32")."""
    dt = 0
    q, k, v = rand_qkv(1, 4, 2, 333, 333, 128, dt, seed=31, kbias=1.0)
    v = (v.float() + 3.0 * torch.randn(1, 2, 1, 128, generator=torch.Generator().manual_seed(1))).half()
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    km = util.bits(sq.channel_mean(kd))
    if pv == "f8":
        o = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_warp", pv_accum_dtype="fp32", smooth_v=True)
        vm = sq.per_channel_fp8(vd, smooth_v=True)[2].cpu().numpy()
    else:
        o = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_warp", pv_accum_dtype="fp16", smooth_v=True)
        vm = sq.channel_mean(vd).float().cpu().numpy()
    torch.cuda.synchronize()
    ref, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv=pv,
                                          qk_quant_gran="per_warp", km=km, smooth_v=True, vm=vm, fp8_scores=SCORES)
    got, ref = o.float().cpu().numpy(), util.f32(ref, dt)
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-3 * scale + util.out_ulp(scale, dt)
    truth = util.sdpa_f32(q, k, v, causal).numpy()
    rel = util.rmse(got, truth) / float(np.sqrt((truth ** 2).mean()))
    REPORT[f"sdpa/smooth_v/{pv}/{'c' if causal else 'nc'}"] = dict(rel_rmse=rel, cos=util.cos_sim(got, truth))
    assert rel <= (0.02 if pv == "f8" else 0.01)


# ------------------------------------------------------------------------------------------------ attention kernel vs oracle
CASES = [
    # name,                    B Hq Hkv  Lq   Lk   D   dt
    ("single_tile_d128",       1, 1, 1, 128,  64, 128, 0),
    ("single_tile_d64",        1, 1, 1, 128,  64,  64, 0),
    ("square_512_d128",        1, 2, 2, 512, 512, 128, 0),
    ("gqa_ragged_d128_bf16",   2, 4, 2, 300, 300, 128, 1),
    ("cross_d64",              2, 2, 1, 200, 333,  64, 0),
    ("long_kv_d128",           1, 2, 1, 130, 1100, 128, 1),
]


PV_ACCUM = {"f8_two": "fp32+fp32", "f8_single": "fp32", "f8f_two": "fp32+fp32", "f8f_single": "fp32", "f16_two": "fp16+fp32", "f16_single": "fp32"}


def _scores_of(pv):
    """f8_*: FP8 PV in the exact score form (the default); f8f_*: the opt-in folded variant (fp8_scores="folded") against the oracle mode that
    mirrors it; None for FP16 PV (one form: exact)."""
    return None if pv.startswith("f16") else ("folded" if pv.startswith("f8f") else "exact")


@pytest.mark.parametrize("pv", ["f8_two", "f8_single", "f8f_two", "f8f_single", "f16_two", "f16_single"])
@pytest.mark.parametrize("gran", ["per_block", "per_warp", "per_thread"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_attention_kernel_vs_oracle(oracle_mod, case, causal, gran, pv):
    name, B, Hq, Hkv, Lq, Lk, D, dt = case
    O = oracle_mod
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=100 + [c[0] for c in CASES].index(name), kbias=1.5)
    fp8 = pv.startswith("f8")
    scores = _scores_of(pv)
    # the K mean is host plumbing (torch, as in the reference): hand the oracle the very same km
    km = util.bits(sq.channel_mean(k.to(DEV)))
    o_bits, lse_ref, aux = O.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal,
                                            pv="f8" if fp8 else "f16", qk_quant_gran=gran, return_lse=True, km=km,
                                            warpq=16 if (pv == "f16_two" and D == 128) else 32,   # core.py:602
                                            fp8_scores=scores or "exact", single_level=pv.endswith("single") and fp8)
    # same operands through the HIP kernels
    fn = sa.sageattn_qk_int8_pv_fp8_cuda if fp8 else sa.sageattn_qk_int8_pv_fp16_cuda
    accum = PV_ACCUM[pv]
    kw = dict(fp8_scores=scores) if fp8 else {}
    o, lse = fn(q.to(DEV), k.to(DEV), v.to(DEV), is_causal=causal, qk_quant_gran=gran, pv_accum_dtype=accum, return_lse=True, **kw)
    torch.cuda.synchronize()
    got, ref = o.float().cpu().numpy(), util.f32(o_bits, dt)
    assert np.isfinite(got).all()
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"kernel_vs_oracle/{name}/{'c' if causal else 'nc'}/{gran}/{pv}"] = dict(max_abs=err, max_o=scale,
                                                                                    lse=float(np.abs(lse.cpu().numpy() - lse_ref).max()))
    out_ulp = util.out_ulp(scale, dt)            # one output-dtype ulp at max|o|
    assert err <= 2e-3 * scale + out_ulp, f"max|diff| {err:.3e} vs max|o| {scale:.3e}"
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3     # q.km^T correction is rounded to fp16/bf16


@pytest.mark.parametrize("gran", ["per_warp", "per_thread"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("case", [CASES[i] for i in (0, 2, 4, 5)], ids=[CASES[i][0] for i in (0, 2, 4, 5)])
def test_sm90_entry_point_scale_groups_vs_oracle(oracle_mod, case, causal, gran):
    """sageattn_qk_int8_pv_fp8_cuda_sm90 with the sm90 kernels' scale groups (core.py:964-970): q scales per 16 rows
    (or the 8 per-thread slots of every 16 rows), k scales per 128 keys -- against the oracle quantising with those groups."""
    name, B, Hq, Hkv, Lq, Lk, D, dt = case
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=300 + Lq, kbias=1.5)
    km = util.bits(sq.channel_mean(k.to(DEV)))
    o_bits, lse_ref, aux = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8",
                                                     qk_quant_gran=gran, return_lse=True, km=km, warpq=16, blkk=128, fp8_scores=SCORES)
    o, lse = sa.sageattn_qk_int8_pv_fp8_cuda_sm90(q.to(DEV), k.to(DEV), v.to(DEV), is_causal=causal, qk_quant_gran=gran,
                                                  pv_accum_dtype="fp32+fp32", return_lse=True)
    torch.cuda.synchronize()
    got, ref = o.float().cpu().numpy(), util.f32(o_bits, dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"sm90_groups/{name}/{'c' if causal else 'nc'}/{gran}"] = dict(max_abs=err, max_o=scale)
    assert np.isfinite(got).all() and err <= 2e-3 * scale + (2 ** -7 if dt == 1 else 2 ** -10) * scale
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3
    # the reference's OWN sm90 kernel iterates over 128 keys (one maximum update and one RO += RO_temp per 128 keys, sm90.cu:285-356); ours takes
    # 64, like the oracle above.  Against the oracle in the 128-key schedule the distance is that of re-rolled e4m3 roundings: a statistical bar
    # (tests/test_oracle_golden.py::test_sm90_tile_schedule_64_vs_128_keys pins the two oracle schedules against each other at the same bar)
    o128, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8", qk_quant_gran=gran, km=km,
                                           warpq=16, blkk=128, fp8_scores=SCORES, tile_keys=128)
    r128 = util.f32(o128, dt)
    rel_rms = float(np.sqrt(((got - r128) ** 2).mean() / (r128 ** 2).mean()))
    REPORT[f"sm90_groups/{name}/{'c' if causal else 'nc'}/{gran}"].update(rel_rms_vs_128_key_schedule=rel_rms, max_abs_vs_128=float(np.abs(got - r128).max()))
    assert rel_rms <= 1.5e-2 and np.abs(got - r128).max() <= 3e-2 * scale
    # the INT8 operands themselves: bit-exact against the oracle's quantiser with the same groups
    q8, qs, k8, ks = (sq.per_warp_int8(q.to(DEV), k.to(DEV), util.from_bits(km, dt, DEV), BLKQ=128, WARPQ=16, BLKK=128) if gran == "per_warp"
                      else sq.per_thread_int8(q.to(DEV), k.to(DEV), util.from_bits(km, dt, DEV), BLKQ=128, WARPQ=16, BLKK=128, WARPK=128))
    assert (q8.cpu().numpy() == aux["q8"]).all() and (k8.cpu().numpy() == aux["k8"]).all()
    assert (qs.cpu().numpy() == aux["qs"]).all() and (ks.cpu().numpy() == aux["ks"]).all()
    with pytest.raises(NotImplementedError):
        sa.sageattn_qk_int8_pv_fp8_cuda_sm90(q.to(DEV), k.to(DEV), v.to(DEV), pv_accum_dtype="fp32")


EDGE = [  # B, Hq, Hkv, Lq, Lk, D
    (1, 1, 1, 1, 1, 64), (1, 3, 3, 5, 3, 128), (1, 6, 3, 127, 63, 64), (3, 1, 1, 129, 65, 128),
    (1, 2, 1, 64, 64, 128), (1, 5, 5, 257, 191, 64), (2, 3, 1, 33, 1000, 128),
]


@pytest.mark.parametrize("shape", EDGE, ids=[f"b{a}h{b}k{c}q{d}l{e}d{f}" for a, b, c, d, e, f in EDGE])
@pytest.mark.parametrize("pv", ["f8", "f16"])
@pytest.mark.parametrize("causal", [False, True])
def test_edge_shapes_vs_oracle(oracle_mod, shape, pv, causal):
    """Tiny / ragged / single-token / B*H not a multiple of 8 / GQA / Lq != Lk (incl. causal with Lq != Lk, which
    the CUDA kernels allow, qk_int_sv_f16_cuda_sm80.cu:218-222, top-left aligned)."""
    B, Hq, Hkv, Lq, Lk, D = shape
    dt = 1 if (Lq + Lk) % 2 else 0
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=Lq * 7 + Lk, kbias=1.0)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    fn = sa.sageattn_qk_int8_pv_fp8_cuda if pv == "f8" else sa.sageattn_qk_int8_pv_fp16_cuda
    o, lse = fn(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32+fp32" if pv == "f8" else "fp32", return_lse=True)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv=pv,
                                                qk_quant_gran="per_thread", return_lse=True, km=km, fp8_scores=SCORES)
    got, ref = o.float().cpu().numpy(), util.f32(ref, dt)
    assert np.isfinite(got).all()
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-3 * scale + util.out_ulp(scale, dt)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3


DEGENERATE = ["k_constant", "v_zero", "q_zero", "tiny", "moderate", "large", "huge", "one_hot_rows"]


def degenerate_qkv(what, D):
    B, Hq, Hkv, L, dt = 1, 4, 2, 330, 0
    q, k, v = rand_qkv(B, Hq, Hkv, L, L, D, dt, seed=61, kbias=1.0)
    if what == "k_constant":
        k = k[:, :, :1].expand(-1, -1, L, -1).contiguous()
    elif what == "v_zero":
        v = torch.zeros_like(v)
    elif what == "q_zero":
        q = torch.zeros_like(q)
    elif what == "tiny":
        q, k, v = (q.float() * 1e-4).half(), (k.float() * 1e-4).half(), (v.float() * 1e-4).half()
    elif what in ("moderate", "large", "huge"):
        f = {"moderate": 8.0, "large": 40.0, "huge": 1000.0}[what]
        q, k, v = (q.float() * f).half(), (k.float() * f).half(), (v.float() * 2000).half()
    else:
        q = q.clone(); q[..., 0] = 30000.0
        k = k.clone(); k[:, :, ::7, 3] = -20000.0
    return q, k, v, dt


@pytest.mark.parametrize("what", DEGENERATE)
@pytest.mark.parametrize("api", ["f8", "f8f", "f16", "triton", "varlen"])
def test_degenerate_inputs_vs_oracle(oracle_mod, api, what):
    """Inputs at the edges of the quantisers.  K constant over the tokens (k - mean == 0: every k scale 0), V == 0 (every FP8 V scale 0), Q == 0, fp16 subnormals
    (x 1e-4), and magnitudes that drive c -- the exponent change per INT8 x INT8 score step, sm_scale log2(e) q_scale k_scale, 1e-4 on randn inputs -- to 5e-3
    (moderate, x 8), 0.14 (large, x 40), 90 (huge, x 1000) and 4000 (rows with one element of 30000: a q scale set by one lane).
    EVERY DEFAULT ROUTE -- f8 (sageattn_qk_int8_pv_fp8_cuda without fp8_scores=), f16, triton and sageattn_varlen -- is held to the usual bar,
    2e-3 * max|o| + one output ulp, against the EXACT oracle (the reference's exp2(fma(s, c, -m)), attn_utils.cuh:445-449) on every one of these
    inputs, outputs and LSE: since round 6 every default route evaluates that formula (rounds 4-5 folded the bias of the score's bit pattern into the
    scale FMA, 1 % ... > 100 % of max|o| off from c ~ 0.1 on).  f8f is the opt-in folded FP8 variant against the oracle mode that mirrors it.
    Inside the key loop P saturates like the reference's cvt.rn.satfinite (MODE.FP16_OVFL): finite outputs everywhere."""
    D = 128 if api not in ("triton", "varlen") else 64
    q, k, v, dt = degenerate_qkv(what, D)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    km = util.bits(sq.channel_mean(kd))
    fp8 = api.startswith("f8")
    form = "folded" if api == "f8f" else ("exact" if api != "f8" else SCORES)      # (SAGE_FP8_SCORES=folded runs of the suite: the process default's own oracle mode)
    for causal in (False, True):
        if api == "triton":
            ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16_triton",
                                                        qk_quant_gran="per_block", return_lse=True, km=km)
            o, lse = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, is_causal=causal, return_lse=True)
        elif api == "varlen":
            # the four heads' tokens as ONE packed batch of B = 1 sequence per call would hide the packed route: two sequences of unequal length
            B, Hq, L = q.shape[0], q.shape[1], q.shape[2]
            cut = 130
            cu = torch.tensor([0, cut, L], dtype=torch.int32)
            qp, kp, vp = (t[0].transpose(0, 1).contiguous() for t in (q, k, v))      # [L, H, D]
            kmp = util.bits(_varlen_km(kp.to(DEV), cu.to(DEV), cu.to(DEV)))
            ref = oracle_mod.sageattn_varlen(util.bits(qp), util.bits(kp), util.bits(vp), dt, cu.numpy(), cu.numpy(), is_causal=causal, km=kmp)
            o = sa.sageattn_varlen(qp.to(DEV), kp.to(DEV), vp.to(DEV), cu.to(DEV), cu.to(DEV), L - cut, L - cut, is_causal=causal)
            lse = lse_ref = None
        else:
            ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8" if fp8 else "f16",
                                                        qk_quant_gran="per_thread", return_lse=True, km=km, fp8_scores=form)
            if fp8:
                kw = dict(fp8_scores="folded") if api == "f8f" else {}          # f8: the default route, no form named
                o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_thread", pv_accum_dtype="fp32+fp32", return_lse=True, **kw)
            else:
                o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_thread", pv_accum_dtype="fp32", return_lse=True)
        torch.cuda.synchronize()
        got, want = o.float().cpu().numpy(), util.f32(ref, dt)
        tag = f"{api} {what} causal={causal}"
        assert np.isfinite(want).all(), f"oracle {tag}"
        assert np.isfinite(got).all(), tag
        scale = float(np.abs(want).max())
        err = float(np.abs(got - want).max())
        REPORT[f"degenerate/{api}/{what}/{'c' if causal else 'nc'}"] = dict(max_abs=err, max_o=scale)
        assert err <= 2e-3 * scale + util.out_ulp(scale, dt), f"{tag}: {err:.3e} vs {scale:.3e}"
        if lse is not None:
            # log2-domain maxima of 1e8 (huge) carry an fp32 ulp of 8: the bar scales with the magnitude of the LSE itself
            lg, lr = lse.cpu().numpy(), lse_ref
            fin = np.isfinite(lr)
            assert (np.isfinite(lg) == fin).all(), tag
            if fin.any():       # (the folded variant on one_hot_rows: every row's LSE is -inf, in the kernel and in the oracle mode that mirrors it)
                assert np.abs(lg[fin] - lr[fin]).max() <= 5e-3 * max(1.0, float(np.abs(lr[fin]).max()) / 64), tag


RISING = ["randn", "ramp", "stairs", "falling", "late_spike", "early_spike"]


def rising_qkv(what, L, D, dt=0):
    """Score profiles along the key axis for the lazily refreshed softmax reference of the FP16-PV loops (the running maximum is refreshed, and O / l
    rescaled, only when a row of the wave exceeds the reference by more than 8 in the log2 domain): randn (one refresh, in the first tile), a ramp of the
    key magnitude x 1 ... x 6 (the maximum creeps up inside the window: P up to 2^8 against a stale reference), stairs (x 4 every 512 keys: jumps
    beyond the window), falling (the first tile holds the maximum, everything later is tiny), and one key 12 x larger than the rest late / early."""
    g = torch.Generator().manual_seed(4242 + L)
    q = torch.randn(1, 4, L, D, generator=g)
    k = torch.randn(1, 2, L, D, generator=g)
    v = torch.randn(1, 2, L, D, generator=g)
    t = torch.arange(L, dtype=torch.float32).view(1, 1, L, 1)
    if what == "ramp":
        k = k * (1.0 + 5.0 * t / L)
    elif what == "stairs":
        k = k * (4.0 ** torch.floor(t / 512.0)).clamp(max=256.0) * 0.25
    elif what == "falling":
        k = k * (6.0 - 5.5 * t / L)
    elif what == "late_spike":
        k[:, :, L - 70] *= 12.0
    elif what == "early_spike":
        k[:, :, 5] *= 12.0
    dtype = torch.float16 if dt == 0 else torch.bfloat16
    return q.to(dtype), k.to(dtype), v.to(dtype)


@pytest.mark.parametrize("what", RISING)
@pytest.mark.parametrize("api", ["f16", "f16_two", "triton", "varlen", "f8"])
def test_score_profiles_along_the_key_axis_vs_oracle(oracle_mod, api, what):
    """2048 + 90 keys (30 pipelined tiles and a ragged tail), causal and not, D = 128 and 64: the FP16-PV routes (lazy reference) and, as a control, the FP8
    default route (reference refreshed whenever a maximum moves) against the exact oracle, which updates m in every tile as the reference's kernels do
    (attn_utils.cuh:394-431) -- outputs at the usual bar, LSE too."""
    L = 2138
    for D, causal in ((128, True), (64, False), (128, False)):
        q, k, v = rising_qkv(what, L, D)
        dt = 0
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        km = util.bits(sq.channel_mean(kd))
        lse = lse_ref = None
        if api == "triton":
            ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16_triton",
                                                        qk_quant_gran="per_block", return_lse=True, km=km)
            o, lse = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, is_causal=causal, return_lse=True)
        elif api == "varlen":
            cut = 700
            cu = torch.tensor([0, cut, L], dtype=torch.int32)
            qp, kp, vp = (t[0].transpose(0, 1).contiguous() for t in (q, k, v))
            kmp = util.bits(_varlen_km(kp.to(DEV), cu.to(DEV), cu.to(DEV)))
            ref = oracle_mod.sageattn_varlen(util.bits(qp), util.bits(kp), util.bits(vp), dt, cu.numpy(), cu.numpy(), is_causal=causal, km=kmp)
            o = sa.sageattn_varlen(qp.to(DEV), kp.to(DEV), vp.to(DEV), cu.to(DEV), cu.to(DEV), L - cut, L - cut, is_causal=causal)
        elif api == "f8":
            ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8",
                                                        qk_quant_gran="per_thread", return_lse=True, km=km, fp8_scores=SCORES)
            o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32+fp32", return_lse=True)
        else:
            two = api == "f16_two"
            ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16",
                                                        qk_quant_gran="per_warp" if two else "per_thread", return_lse=True, km=km,
                                                        warpq=16 if (two and D == 128) else 32)
            o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_warp" if two else "per_thread",
                                                      pv_accum_dtype="fp16+fp32" if two else "fp32", return_lse=True)
        torch.cuda.synchronize()
        got, want = o.float().cpu().numpy(), util.f32(ref, dt)
        tag = f"{api} {what} D={D} causal={causal}"
        assert np.isfinite(got).all(), tag
        scale = float(np.abs(want).max())
        err = float(np.abs(got - want).max())
        REPORT[f"score_profiles/{api}/{what}/d{D}/{'c' if causal else 'nc'}"] = dict(max_abs=err, max_o=scale)
        assert err <= 2e-3 * scale + util.out_ulp(scale, dt), f"{tag}: {err:.3e} vs {scale:.3e}"
        if lse is not None:
            assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3 * max(1.0, float(np.abs(lse_ref).max()) / 64), tag


VROWS_SHAPES = [   # B, Hq, Hkv, Lq, Lk, D
    (2, 4, 2, 300, 300, 128), (1, 3, 3, 129, 1000, 64), (1, 8, 8, 1024, 1024, 128), (2, 2, 1, 5, 70, 128), (1, 2, 2, 1, 1, 64),
    (1, 4, 4, 2138, 2138, 128), (1, 6, 2, 640, 640, 64), (1, 2, 1, 200, 63, 128),
]


@pytest.mark.parametrize("shape", VROWS_SHAPES, ids=[f"b{a}h{b}k{c}q{d}l{e}d{f}" for a, b, c, d, e, f in VROWS_SHAPES])
@pytest.mark.parametrize("layout", ["HND", "NHD"])
@pytest.mark.parametrize("api", ["cuda", "triton"])
def test_v_rows_in_place_is_bit_identical_to_the_tile_image_route(api, layout, shape):
    """fp16 inputs, FP16 PV, fused Q quantisation: the attention kernel reads V's rows in place (LDS-DMA of rows + transposing LDS reads,
    sage_attn_fused_q*_pv_f16_vrows) instead of a pre-transposed tile image.  Same operands into the same MFMAs: every output bit and every
    LSE bit equal to the image route's, causal and not, both layouts, GQA, ragged and tiny lengths, a strided V view; and the route is really
    taken by default (the vrows entry point is what gets called, the V image pass is not)."""
    B, Hq, Hkv, Lq, Lk, D = shape
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, 0, seed=500 + Lq + Lk, kbias=1.0)
    qd, kd, v0 = to_dev(q, layout), to_dev(k, layout), to_dev(v, layout)
    # V as a view into a wider buffer (every stride doubled): rows are 16-byte aligned but not contiguous
    wide = torch.zeros(*v0.shape[:-1], 2 * D, dtype=v0.dtype, device=DEV)
    wide[..., D:] = v0
    vd = wide[..., D:]
    fn = sa.sageattn_qk_int8_pv_fp16_cuda if api == "cuda" else sa.sageattn_qk_int8_pv_fp16_triton
    lib = _cabi.load()
    name = "sage_attn_fused_q_pv_f16_vrows" if api == "cuda" else "sage_attn_fused_qblock_pv_f16_vrows"
    real, calls = getattr(lib, name), []

    def counting(*a):
        calls.append(1)
        return real(*a)
    for causal in ((False, True) if Lq == Lk else (False,)):
        o_img, lse_img = fn(qd, kd, vd, tensor_layout=layout, is_causal=causal, return_lse=True, v_in_place=False)
        setattr(lib, name, counting)
        try:
            o_row, lse_row = fn(qd, kd, vd, tensor_layout=layout, is_causal=causal, return_lse=True)         # the default route (these sizes: rows)
        finally:
            setattr(lib, name, real)
        torch.cuda.synchronize()
        assert len(calls) >= 1, "fp16 inputs take the V-rows route by default"
        assert torch.equal(o_row, o_img), f"{api} {layout} {shape} causal={causal}: {(o_row.float() - o_img.float()).abs().max().item()}"
        assert torch.equal(lse_row, lse_img)
        calls.clear()
    # bf16 inputs need the conversion pass: the image route, never the rows entry point
    setattr(lib, name, counting)
    try:
        fn(qd.bfloat16(), kd.bfloat16(), vd.bfloat16(), tensor_layout=layout)
    finally:
        setattr(lib, name, real)
    assert not calls


@pytest.mark.parametrize("gran,accum", [("per_block", "triton"), ("per_block", "cuda"), ("per_warp", "cuda"), ("per_thread", "cuda")])
@pytest.mark.parametrize("shape", VROWS_SHAPES, ids=[f"b{a}h{b}k{c}q{d}l{e}d{f}" for a, b, c, d, e, f in VROWS_SHAPES])
@pytest.mark.parametrize("layout,causal", [("HND", False), ("NHD", True)])
def test_int8_q_v_rows_in_place_is_bit_identical_to_the_tile_image(shape, layout, causal, gran, accum):
    """ABI 21, ``sage_attn_qk_int8_pv_f16_vrows``: the reference's native FP16-PV op signature (INT8 q / k, their scales, the fp16 value tensor
    as it is) against the image entry point on the image of the same V -- same operands into the same MFMAs, bit for bit, in every scale
    grouping and both kernel forms."""
    B, Hq, Hkv, Lq, Lk, D = shape
    if causal and Lq != Lk:
        pytest.skip("causal needs Lq == Lk")
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, 0, seed=Lq + Lk + D, kbias=1.0)
    q, k, v = to_dev(q, layout), to_dev(k, layout), to_dev(v, layout)
    km = sq.channel_mean(k, layout)
    km4 = km.unsqueeze(1 if layout == "NHD" else 2)
    if gran == "per_block":
        q8, qs, k8, ks = sq.per_block_int8(q, k, km=km4, sm_scale=D ** -0.5, tensor_layout=layout)
        g, warp, sml2 = _cabi.GRAN_PER_BLOCK, 128, 1.0
    elif gran == "per_warp":
        q8, qs, k8, ks = sq.per_warp_int8(q, k, km=km4, tensor_layout=layout)
        g, warp, sml2 = _cabi.GRAN_PER_WARP, 32, D ** -0.5 * sq.LOG2E
    else:
        q8, qs, k8, ks = sq.per_thread_int8(q, k, km=km4, tensor_layout=layout)
        g, warp, sml2 = _cabi.GRAN_PER_THREAD, 32, D ** -0.5 * sq.LOG2E
    acc = _cabi.PV_ACCUM_TRITON if accum == "triton" else _cabi.PV_ACCUM_SINGLE
    lay = 0 if layout == "NHD" else 1
    outs = []
    for vv in (sq.prep_v_fp16(v, layout), v):
        o = torch.empty(q.shape, dtype=torch.float16, device=DEV)
        lse = sa_ops.qk_int8_sv_f16_attn_impl(q8, k8, vv, o, qs, ks, None, lay, int(causal), g, warp, sml2, acc, 1)
        torch.cuda.synchronize()
        outs.append((o, lse))
    assert torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name", ["c1_b1h4n512d64_f16", "gqa_causal_n300d128_bf16", "cross_lq200_lk333_d64_f16", "causal_n384d128_f16",
                                  "long_nc_lq256_lk1100_d128_f16", "long_c_n1000_d64_bf16"])
def test_kernel_level_forward_on_the_reference_s_own_int8_operands(name):
    """``sageattention.triton.attn_qk_int8_per_block[_causal].forward`` -- the reference's kernel-level entry (what its bench script times) -- fed the
    INT8 tensors and scales the REFERENCE's quantiser produced (in the fixture), against the reference kernel's own output: the attention
    kernel alone, pinned to a reference output with nothing of ours in front of it."""
    from sageattention.triton.attn_qk_int8_per_block import forward
    from sageattention.triton.attn_qk_int8_per_block_causal import forward as forward_causal
    z, (B, Hq, Hkv, Lq, Lk, D, dt, causal) = util.golden(name)
    q8, k8 = torch.from_numpy(z["q_int8"]).to(DEV), torch.from_numpy(z["k_int8"]).to(DEV)
    qs, ks = torch.from_numpy(z["q_scale"]).to(DEV), torch.from_numpy(z["k_scale"]).to(DEV)
    v = util.from_bits(z["v"], dt, DEV).to(torch.float16)            # core.py:297-298
    odt = torch.float16 if dt == 0 else torch.bfloat16
    o, lse = (forward_causal if causal else forward)(q8, k8, v, qs, ks, output_dtype=odt, return_lse=True)
    torch.cuda.synchronize()
    assert o.dtype == odt and o.shape == q8.shape and lse.shape == (B, Hq, Lq)
    got, ref = o.float().cpu().numpy(), util.f32(z["o"], dt)
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-3 * scale + (2 ** -7 * scale if dt == 1 else 0.0)
    o2, lse2 = (forward_causal if causal else forward)(q8, k8, v, qs.half().unsqueeze(-1), ks.half().unsqueeze(-1), output_dtype=odt)   # (scale tensors as the reference's bench script hands them)
    assert lse2.numel() == 0 and o2.shape == o.shape


@pytest.mark.parametrize("name", ["varlen_nc_d64_f16", "varlen_c_d64_f16", "varlen_c_d128_bf16", "varlenx_nc_d128_bf16", "varlenx_c_d64_f16"])
def test_kernel_level_varlen_forward_on_the_reference_s_own_int8_operands(name):
    """The packed counterparts (attn_qk_int8_block_varlen.forward / attn_qk_int8_per_block_causal_varlen.forward) on the reference quantiser's tensors."""
    from sageattention.triton.attn_qk_int8_block_varlen import forward
    from sageattention.triton.attn_qk_int8_per_block_causal_varlen import forward as forward_causal
    z, meta = util.golden(name)
    dt, causal, D = meta[6], meta[7], meta[5]
    cu_q = torch.from_numpy(z["cu_q"] if "cu_q" in z.files else z["cu"]).to(torch.int32).to(DEV)
    cu_k = torch.from_numpy(z["cu_k"] if "cu_k" in z.files else z["cu"]).to(torch.int32).to(DEV)
    q8, k8 = torch.from_numpy(z["q_int8"]).to(DEV), torch.from_numpy(z["k_int8"]).to(DEV)
    qs, ks = torch.from_numpy(z["q_scale"]).to(DEV), torch.from_numpy(z["k_scale"]).to(DEV)
    cu_qs, cu_ks = torch.from_numpy(z["cu_qs"]).to(torch.int32).to(DEV), torch.from_numpy(z["cu_ks"]).to(torch.int32).to(DEV)
    v = util.from_bits(z["v"], dt, DEV).to(torch.float16)
    max_q = int((cu_q[1:] - cu_q[:-1]).max().item())
    odt = torch.float16 if dt == 0 else torch.bfloat16
    o = (forward_causal if causal else forward)(q8, k8, v, cu_q, cu_k, max_q, qs, ks, cu_qs, cu_ks, output_dtype=odt)
    torch.cuda.synchronize()
    got, ref = o.float().cpu().numpy(), util.f32(z["o"], dt)
    scale = float(np.abs(ref).max())
    assert np.abs(got - ref).max() <= 2e-3 * scale + (2 ** -7 * scale if dt == 1 else 0.0)


def test_reference_module_names_import_and_quantise_like_the_reference():
    """``sageattention.core`` / ``.quant`` / ``.triton.*``: the reference's module paths resolve, with its signatures; the quantisers behind them return
    the reference's bits (the per-block fixture) and the CUDA-convention one differs from the Triton one only in rounding."""
    import sageattention.core as rc
    import sageattention.quant as rq
    from sageattention.triton.quant_per_block import per_block_int8
    from sageattention.triton.quant_per_thread import per_thread_int8
    from sageattention.triton.quant_per_block_varlen import per_block_int8 as per_block_int8_varlen
    assert rc.sageattn is sa.sageattn and rc.sageattn_varlen is sa.sageattn_varlen
    z, (B, Hq, Hkv, Lq, Lk, D, dt, causal) = util.golden("c1_b1h4n512d64_f16")
    q, k = util.from_bits(z["q"], dt, DEV), util.from_bits(z["k"], dt, DEV)
    km = util.from_bits(z["km"], dt, DEV)
    q8, qs, k8, ks = per_block_int8(q, k, km=km, sm_scale=D ** -0.5)
    assert (q8.cpu().numpy() == z["q_int8"]).all() and (k8.cpu().numpy() == z["k_int8"]).all()
    assert (qs.cpu().numpy() == z["q_scale"]).all() and (ks.cpu().numpy() == z["k_scale"]).all()
    q8c, qsc, k8c, ksc = rq.per_block_int8(q, k, km=km, sm_scale=D ** -0.5)
    assert (q8c.int() - q8.int()).abs().max().item() <= 1 and q8c.shape == q8.shape
    assert rq.per_warp_int8(q, k, km=km)[1].shape == (B, Hq, 4 * ((Lq + 127) // 128))
    assert per_thread_int8(q, k, km=km)[3].shape == (B, Hkv, 4 * ((Lk + 63) // 64))
    zv, _ = util.golden("varlen_c_d64_f16")
    qv, kv = util.from_bits(zv["q"], 0, DEV), util.from_bits(zv["k"], 0, DEV)
    cu = torch.from_numpy(zv["cu"]).to(torch.int32).to(DEV)
    kv = kv - kv.mean(dim=0, keepdim=True)                              # core.py:432-434
    r = per_block_int8_varlen(qv, kv, cu, cu, 257, 257, sm_scale=64 ** -0.5)
    assert len(r) == 6 and (r[0].cpu().numpy() == zv["q_int8"]).all() and (r[2].cpu().numpy() == zv["k_int8"]).all()
    assert (r[4].cpu().numpy() == zv["cu_qs"]).all() and (r[5].cpu().numpy() == zv["cu_ks"]).all()


def test_strided_views_of_a_packed_qkv_tensor():
    """q/k/v as non-contiguous views (the usual fused-QKV projection output): strides are honoured, no copies."""
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(2, 300, 3, 8, 128, generator=g).half().to(DEV)          # [B, L, 3, H, D]
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                          # NHD views, stride(1) = 3*H*D
    o = sa.sageattn(q, k, v, tensor_layout="NHD", is_causal=True)
    o_ref = sa.sageattn(q.contiguous(), k.contiguous(), v.contiguous(), tensor_layout="NHD", is_causal=True)
    torch.cuda.synchronize()
    assert o.shape == q.shape and torch.equal(o, o_ref)
    oh = sa.sageattn(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), tensor_layout="HND", is_causal=True)
    assert torch.equal(oh.transpose(1, 2), o_ref)


def test_long_sequence_32k_vs_oracle(oracle_mod):
    """N = 32768 (the longest point of the published sweep), one GQA group, against the CPU oracle."""
    q, k, v = rand_qkv(1, 2, 1, 32768, 32768, 128, 1, seed=77, kbias=1.0)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o = sa.sageattn(qd, kd, vd, is_causal=True)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 1, is_causal=True, pv="f8",
                                          qk_quant_gran="per_thread", km=km, fp8_scores=SCORES)
    got, ref = o.float().cpu().numpy(), util.f32(ref, 1)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT["kernel_vs_oracle/n32768_causal_f8"] = dict(max_abs=err, max_o=scale)
    assert err <= 2e-3 * scale + 2 ** -8 * scale


# ------------------------------------------------------------------------------------------------ FP8 score forms (the one rule)
FORM_CASES = [
    # name,                       B  Hq Hkv  Lq    Lk    D   dt causal kbias
    ("c3_like_n2048_d128_causal", 1, 8, 8, 2048, 2048, 128, 1, True, 0.0),
    ("c5_like_n4400_d64",         1, 6, 6, 4400, 4400,  64, 1, False, 0.0),
    ("biased_k_n1024_d128",       2, 4, 2, 1024, 1024, 128, 0, True, 5.0),
    ("short_n300_d128_causal",    2, 4, 4,  300,  300, 128, 0, True, 1.0),
    ("short_n512_d64",            2, 4, 2,  512,  512,  64, 0, False, 1.0),
    ("short_n333_d64_cross",      2, 4, 4,  200,  333,  64, 1, False, 2.0),
]


@pytest.mark.parametrize("case", FORM_CASES, ids=[c[0] for c in FORM_CASES])
def test_fp8_score_forms_exact_default_and_folded_variant(oracle_mod, case):
    """The default FP8 route is the EXACT score form -- the one pinned to the reference's formula -- and meets the exact oracle at 2e-3 * max|o|.
    The folded form is an opt-in variant (fp8_scores="folded") described by three measured clauses: (i) kernel(folded) meets
    oracle(folded), the oracle mode that mirrors it rounding for rounding, at 2e-3 * max|o|; (ii) folded against exact: rel-RMS <= 1e-2 on
    these ordinary-magnitude inputs (kernel vs kernel here, oracle vs oracle on the CPU in tests/test_oracle_golden.py); (iii) cos / rel-RMSE vs
    fp32 SDPA within 1e-4 / 1e-3 of the exact form's."""
    name, B, Hq, Hkv, Lq, Lk, D, dt, causal, kbias = case
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=900 + Lq, kbias=kbias)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    out = {}
    for form in ("folded", "exact"):
        out[form] = sa.sageattn(qd, kd, vd, is_causal=causal, fp8_scores=form)
    o_default = sa.sageattn(qd, kd, vd, is_causal=causal)
    torch.cuda.synchronize()
    assert torch.equal(o_default, out[SCORES]), "the default form is the one SAGE_FP8_SCORES names (exact unless set)"
    km = util.bits(sq.channel_mean(kd))
    truth = util.sdpa_f32(q, k, v, causal).numpy()
    tn = float(np.sqrt((truth ** 2).mean()))
    stats = {}
    for form in ("folded", "exact"):
        got = out[form].float().cpu().numpy()
        ref, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8",
                                              qk_quant_gran="per_thread", km=km, fp8_scores=form)
        _assert_vs_oracle(f"score_forms/{name}/{form}", got, ref, dt)                                       # (i)
        stats[form] = (util.cos_sim(got, truth), util.rmse(got, truth) / tn)
    a, b = out["folded"].float().cpu().numpy(), out["exact"].float().cpu().numpy()
    rel_rms = float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
    REPORT[f"score_forms/{name}"] = dict(rel_rms_folded_vs_exact=rel_rms, max_abs=float(np.abs(a - b).max()), max_o=float(np.abs(b).max()),
                                         cos_folded=stats["folded"][0], cos_exact=stats["exact"][0],
                                         rel_rmse_folded=stats["folded"][1], rel_rmse_exact=stats["exact"][1])
    assert rel_rms <= 1e-2, rel_rms                                                                          # (ii)
    assert abs(stats["folded"][0] - stats["exact"][0]) <= 1e-4 and abs(stats["folded"][1] - stats["exact"][1]) <= 1e-3, stats   # (iii)


# ------------------------------------------------------------------------------------------------ golden (reference outputs)
@pytest.mark.parametrize("name", ["c1_b1h4n512d64_f16", "gqa_causal_n300d128_bf16", "cross_lq200_lk333_d64_f16",
                                  "causal_n384d128_f16", "pad_d96_n160_f16", "long_nc_lq256_lk1100_d128_f16", "long_c_n1000_d64_bf16"])
@pytest.mark.parametrize("layout", ["HND", "NHD"])
def test_triton_api_vs_reference_golden(name, layout):
    z, (B, Hq, Hkv, Lq, Lk, D, dt, causal) = util.golden(name)
    q, k, v = (to_dev(util.from_bits(z[n], dt), layout) for n in ("q", "k", "v"))
    o, lse = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, is_causal=bool(causal), return_lse=True)
    torch.cuda.synchronize()
    assert o.shape == q.shape and o.dtype == q.dtype
    got = to_hnd(o, layout).float().cpu().numpy()
    ref = util.f32(z["o"], dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"golden/{name}/{layout}"] = dict(max_abs=err, max_o=scale, lse=float(np.abs(lse.cpu().numpy() - z["lse"]).max()))
    tol = 2e-3 * scale + (2 ** -7 * scale if dt == 1 else 0.0)
    assert err <= tol
    assert np.abs(lse.cpu().numpy() - z["lse"]).max() <= 2e-3
    # the quantised operands themselves are bit-exact vs the reference's
    if D in (64, 128):
        km = util.from_bits(z["km"], dt, DEV)
        q8, qs, k8, ks = sq.per_block_int8(to_dev(util.from_bits(z["q"], dt), "HND"), to_dev(util.from_bits(z["k"], dt), "HND"),
                                           km=km, sm_scale=D ** -0.5)
        assert (q8.cpu().numpy() == z["q_int8"]).all() and (k8.cpu().numpy() == z["k_int8"]).all()
        assert (qs.cpu().numpy() == z["q_scale"]).all() and (ks.cpu().numpy() == z["k_scale"]).all()


@pytest.mark.parametrize("name,kind", [("mask_bool_lq300_lk333_d64_f16", "bool"), ("mask_add_lq200_lk256_d128_bf16", "add"),
                                       ("mask_bool_skipall_lq140_lk130_d64_f16", "bool")])
@pytest.mark.parametrize("layout", ["HND", "NHD"])
def test_attn_mask_vs_reference_golden(name, kind, layout):
    """sageattn_qk_int8_pv_fp16_triton(attn_mask=...) against the reference Triton kernel's output, incl. the
    all-False-tile skip, fully masked rows and a broadcast (stride-0) mask."""
    z, (B, Hq, Hkv, Lq, Lk, D, dt, _) = util.golden(name)
    q, k, v = (to_dev(util.from_bits(z[n], dt), layout) for n in ("q", "k", "v"))
    mask = torch.from_numpy(z["mask"]).to(DEV) if kind == "bool" else util.from_bits(z["mask"], dt, DEV)
    o, lse = sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, attn_mask=mask, return_lse=True)
    torch.cuda.synchronize()
    got, ref = to_hnd(o, layout).float().cpu().numpy(), util.f32(z["o"], dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"golden/{name}/{layout}"] = dict(max_abs=err, max_o=scale)
    assert np.isfinite(got).all()
    assert err <= 2e-3 * scale + (2 ** -7 * scale if dt == 1 else 0.0)
    # fully masked rows carry an LSE of about -1e6/log2e where one fp32 ulp is 0.0625
    # (skipall: a query block whose every tile is skipped -- rows of zeros with an lse of -inf, in the reference and here)
    lse, fin = lse.cpu().numpy(), np.isfinite(z["lse"])
    assert np.array_equal(np.isneginf(lse), ~fin)
    assert (np.abs(lse[fin] - z["lse"][fin]) <= 2e-2 + 2e-7 * np.abs(z["lse"][fin])).all()
    with pytest.raises(AssertionError):
        sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, tensor_layout=layout, attn_mask=mask, is_causal=True)


@pytest.mark.parametrize("name", ["varlen_nc_d64_f16", "varlen_c_d64_f16", "varlen_c_d128_bf16"])
def test_varlen_vs_reference_golden(name):
    z, (nseq, Hq, Hkv, total, _, D, dt, causal) = util.golden(name)
    q, k, v = (util.from_bits(z[n], dt, DEV) for n in ("q", "k", "v"))
    cu = torch.from_numpy(z["cu"]).to(DEV)
    lens = np.diff(z["cu"])
    o = sa.sageattn_varlen(q, k, v, cu, cu, int(lens.max()), int(lens.max()), is_causal=bool(causal))
    torch.cuda.synchronize()
    got, ref = o.float().cpu().numpy(), util.f32(z["o"], dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"golden/{name}"] = dict(max_abs=err, max_o=scale)
    assert err <= 2e-3 * scale + (2 ** -7 * scale if dt == 1 else 0.0)
    # bit-exact quantisation vs the reference's varlen quantiser (km subtraction fused in-kernel)
    km = util.from_bits(oracle_km_packed(z["k"], dt), dt, DEV)              # the reference's torch mean, restated
    q8, qs, k8, ks, cu_qs, cu_ks = sq.per_block_int8_varlen(q, k, cu, cu, int(lens.max()), int(lens.max()), km=km, sm_scale=D ** -0.5)
    assert (cu_qs.cpu().numpy() == z["cu_qs"]).all() and (cu_ks.cpu().numpy() == z["cu_ks"]).all()
    assert (q8.cpu().numpy() == z["q_int8"]).all() and (qs.cpu().numpy() == z["q_scale"]).all()
    assert (k8.cpu().numpy() == z["k_int8"]).all() and (ks.cpu().numpy() == z["k_scale"]).all()


# ------------------------------------------------------------------------------------------------ API behaviour
def test_sageattn_dropin_kwargs_and_accuracy():
    """`F.scaled_dot_product_attention = sageattn`: SDPA-style extras are swallowed (core.py:79-88)."""
    q, k, v = (t.to(DEV) for t in rand_qkv(2, 8, 8, 1024, 1024, 128, 1, seed=7, kbias=3.0))
    o = sa.sageattn(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=True, scale=0.123)
    torch.cuda.synchronize()
    truth = util.sdpa_f32(q, k, v, True).cpu().numpy()
    got = o.float().cpu().numpy()
    cos, rel = util.cos_sim(got, truth), util.rmse(got, truth) / float(np.sqrt((truth ** 2).mean()))
    REPORT["sdpa/sageattn_default_1024"] = dict(cos=cos, rel_rmse=rel)
    assert cos >= 0.999 and rel <= 0.05


@pytest.mark.parametrize("fn_name,kw,cos_min,rel_max", [
    ("sageattn_qk_int8_pv_fp16_triton", {}, 0.9995, 0.02),
    ("sageattn_qk_int8_pv_fp16_cuda", {"pv_accum_dtype": "fp32"}, 0.9995, 0.02),
    ("sageattn_qk_int8_pv_fp8_cuda", {"pv_accum_dtype": "fp32+fp32"}, 0.999, 0.05),
    ("sageattn_qk_int8_pv_fp8_cuda", {"pv_accum_dtype": "fp32+fp16", "qk_quant_gran": "per_warp"}, 0.999, 0.05),
    ("sageattn_qk_int8_pv_fp8_cuda_sm90", {}, 0.999, 0.05),
])
@pytest.mark.parametrize("causal", [False, True])
def test_api_accuracy_vs_sdpa(fn_name, kw, cos_min, rel_max, causal):
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 8, 2, 777, 777, 128, 0, seed=9, kbias=4.0))
    o = getattr(sa, fn_name)(q, k, v, is_causal=causal, **kw)
    torch.cuda.synchronize()
    truth = util.sdpa_f32(q, k, v, causal).cpu().numpy()
    got = o.float().cpu().numpy()
    cos, rel = util.cos_sim(got, truth), util.rmse(got, truth) / float(np.sqrt((truth ** 2).mean()))
    REPORT[f"sdpa/{fn_name}/{kw.get('pv_accum_dtype','default')}/{'c' if causal else 'nc'}"] = dict(cos=cos, rel_rmse=rel, rmse=util.rmse(got, truth))
    assert cos >= cos_min and rel <= rel_max
    assert util.rmse(got, truth) <= 5e-3            # RMSE <= 5e-3 on randn-valued inputs (absolute), both PV precisions


def test_lse_matches_fp32_logsumexp():
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 4, 4, 500, 500, 64, 0, seed=13, kbias=2.0))
    for fn in (sa.sageattn_qk_int8_pv_fp8_cuda, sa.sageattn_qk_int8_pv_fp16_cuda, sa.sageattn_qk_int8_pv_fp16_triton):
        _, lse = fn(q, k, v, is_causal=False, return_lse=True)
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * (64 ** -0.5)
        assert lse.shape == (1, 4, 500) and lse.dtype == torch.float32
        assert (lse - torch.logsumexp(s, dim=-1)).abs().max().item() < 0.05


def test_errors_match_reference_contract():
    q = torch.zeros(1, 2, 64, 160, dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError, match="Unsupported head_dim"):
        sa.sageattn(q, q, q)
    q32 = torch.zeros(1, 2, 64, 64, dtype=torch.float32, device=DEV)
    with pytest.raises(AssertionError):
        sa.sageattn_qk_int8_pv_fp8_cuda(q32, q32, q32)
    qh = torch.zeros(1, 3, 64, 64, dtype=torch.float16, device=DEV)
    kh = torch.zeros(1, 2, 64, 64, dtype=torch.float16, device=DEV)
    with pytest.raises((AssertionError, ValueError)):
        sa.sageattn(qh, kh, kh)


def test_non_default_stream_and_reentrancy():
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 4, 4, 384, 384, 128, 0, seed=21))
    ref = sa.sageattn(q, k, v, is_causal=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        outs = [sa.sageattn(q, k, v, is_causal=True) for _ in range(3)]
    s.synchronize()
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)          # deterministic: same bits on any stream, any repetition


def test_torch_compile_traces_through_the_custom_ops():
    """README.md:30 of the reference: torch.compile in non-fullgraph mode; the fused ops are custom_ops with fakes."""
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 4, 4, 256, 256, 64, 0, seed=17))

    def block(q, k, v):
        return sa.sageattn(q * 1.0, k, v, is_causal=True) + 1.0

    want = block(q, k, v)
    got = torch.compile(block, backend="aot_eager")(q, k, v)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_torch_compile_fullgraph_no_graph_break(backend):
    """The whole dense call is ONE opaque op under torch.compile (ops.sageattn_call): fullgraph=True compiles -- no graph
    break around the ctypes pre-pass -- and the compiled attention is bit-equal to the eager one (same kernels, same route)."""
    q, k, v = (t.to(DEV) for t in rand_qkv(2, 4, 2, 300, 333, 128, 1, seed=23))
    torch._dynamo.reset()

    def attn_only(q, k, v):
        return sa.sageattn(q, k, v, is_causal=False)

    def with_lse(q, k, v):
        o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, pv_accum_dtype="fp32", return_lse=True)
        return o * 2.0, lse

    want = attn_only(q, k, v)
    got = torch.compile(attn_only, backend=backend, fullgraph=True)(q, k, v)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    qs, ks, vs = q[:, :, :256], k[:, :2, :256], v[:, :2, :256]
    o_w, lse_w = with_lse(qs, ks, vs)
    o_g, lse_g = torch.compile(with_lse, backend=backend, fullgraph=True)(qs, ks, vs)
    torch.cuda.synchronize()
    assert torch.equal(lse_g, lse_w) and torch.allclose(o_g.float(), o_w.float(), rtol=1e-2, atol=1e-3)
    from sageattention_amd import ops as ops_mod
    torch.library.opcheck(ops_mod.sageattn_call, (q, k, v, "fp8", "HND", False, "per_thread", None, "fp32+fp32", True, False, True))


# ------------------------------------------------------------------------------------------------ BASELINE.json full sizes
def _props(fn, q, k, v, causal, tag, cos_min, rel_max):
    o_full = fn(q, k, v, is_causal=causal)
    full = fn
    fn = lambda *a, **kw: full(*a, smooth_k=False, **kw)   # invariants below must not depend on torch's k.mean
    o = fn(q, k, v, is_causal=causal)
    # (1) exact linearity in V under power-of-two scaling (per-channel V scales absorb it)
    o2 = fn(q, k, v * 2, is_causal=causal)
    REPORT[f"full/{tag}/v2_maxdiff"] = float((o2.float() - 2 * o.float()).abs().max())
    # exact wherever the output is a normal number of its dtype; in the subnormal range the final
    # rounding has a fixed quantum (2^-24 for fp16), so round(2x) may differ from 2*round(x) by one
    tiny = 2.0 ** -13 if o.dtype == torch.float16 else 2.0 ** -120
    normal = o.abs() >= tiny
    bad = (o2 != o * 2) & normal
    if bool(bad.any()):       # say where: a race shows up as whole rows / tiles, a rounding issue as scattered single elements
        idx = bad.nonzero()
        where = {f"dim{d}": (int(idx[:, d].min()), int(idx[:, d].max()), int(idx[:, d].unique().numel())) for d in range(idx.size(1))}
        raise AssertionError(f"V -> 2V must double the output bit-exactly: {int(bad.sum())} elements differ, index ranges (min, max, distinct) {where}, "
                             f"first {idx[0].tolist()}: {o2[tuple(idx[0])].item()} vs {2 * o[tuple(idx[0])].item()}, nan={bool(o2.isnan().any())}/{bool(o.isnan().any())}")
    assert (o2.float() - 2 * o.float()).abs().max().item() <= 2.0 ** -23
    # (2) batch*head shard invariance: a slice of the heads gives the same bits (multi-GPU sharding)
    hs = slice(q.size(1) // 2, q.size(1) // 2 + 4)
    o_sh = fn(q[1:2, hs], k[1:2, hs], v[1:2, hs], is_causal=causal)
    assert torch.equal(o_sh, o[1:2, hs]), "sharded heads must reproduce the full result bit-exactly"
    # (3) accuracy vs fp32 SDPA on a subset of heads (full fp32 SDPA at this size is the slow part)
    truth = util.sdpa_f32(q[:1, :2], k[:1, :2], v[:1, :2], causal).cpu().numpy()
    got = o_full[:1, :2].float().cpu().numpy()
    cos, rel = util.cos_sim(got, truth), util.rmse(got, truth) / float(np.sqrt((truth ** 2).mean()))
    REPORT[f"full/{tag}"] = dict(cos=cos, rel_rmse=rel, rmse=util.rmse(got, truth))
    assert cos >= cos_min and rel <= rel_max
    assert util.rmse(got, truth) <= 5e-3            # SURVEY 8c's absolute RMSE bar (randn inputs), as written, next to the relative one
    assert torch.isfinite(o_full.float()).all()


def test_config2_fp16_pv_b2h32n4096d128_causal():
    """BASELINE.json configs[1]."""
    q, k, v = (t.to(DEV) for t in rand_qkv(2, 32, 32, 4096, 4096, 128, 0, seed=2))
    # keep V out of the fp16 subnormal range: v_mfma_f32_32x32x16_f16 flushes subnormal fp16 inputs,
    # so a subnormal v (|v| < 2^-14, ~5e-5 of randn samples) contributes 0 while 2v contributes 2v
    v = torch.where(v.abs() < 2.0 ** -12, torch.full_like(v, 2.0 ** -12), v)
    _props(lambda *a, **kw: sa.sageattn_qk_int8_pv_fp16_cuda(*a, pv_accum_dtype="fp32", **kw), q, k, v, True, "c2_f16", 0.9995, 0.02)


def test_config3_fp8_pv_b2h32n8192d128_causal():
    """BASELINE.json configs[2] -- the headline configuration."""
    q, k, v = (t.to(DEV) for t in rand_qkv(2, 32, 32, 8192, 8192, 128, 1, seed=3))
    _props(lambda *a, **kw: sa.sageattn_qk_int8_pv_fp8_cuda(*a, pv_accum_dtype="fp32+fp32", **kw), q, k, v, True, "c3_f8", 0.999, 0.05)


def test_config4_varlen_gqa():
    """BASELINE.json configs[3]: Hq=32, Hkv=8, D=128, mixed lengths 256..16384."""
    lens = [256, 512, 1000, 1024, 2048, 4096, 8192, 16384]
    total = sum(lens)
    g = torch.Generator().manual_seed(4)
    q = torch.randn(total, 32, 128, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(DEV)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    for causal in (False, True):
        o = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
        assert torch.isfinite(o.float()).all()
        # each sequence on its own (batch of one) must give the same bits as inside the packed batch,
        # provided it sees the same K mean: check sequences 2 (ragged 1000) and 0 via smooth_k=False
        o_ns = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal, smooth_k=False)
        for i in (0, 2):
            s, e = int(cu[i]), int(cu[i + 1])
            cu1 = torch.tensor([0, e - s], dtype=torch.int32, device=DEV)
            o1 = sa.sageattn_varlen(q[s:e], k[s:e], v[s:e], cu1, cu1, e - s, e - s, is_causal=causal, smooth_k=False)
            assert torch.equal(o1, o_ns[s:e])
            truth = util.sdpa_f32(q[s:e].transpose(0, 1)[None], k[s:e].transpose(0, 1)[None], v[s:e].transpose(0, 1)[None], causal)
            got = o[s:e].transpose(0, 1)[None].float().cpu().numpy()
            cos = util.cos_sim(got, truth.cpu().numpy())
            REPORT[f"full/c4_varlen/{'c' if causal else 'nc'}/seq{i}"] = dict(cos=cos)
            assert cos >= 0.9995


# ------------------------------------------------------------------------------------------------ BASELINE.json full sizes vs the ORACLE
def _assert_vs_oracle(tag, got, ref_bits, dt, tol_rel=2e-3):
    """max|o_hip - o_oracle| <= 2e-3 * max|o| + one output ulp at max|o| (the bar of the small-shape kernel tests).
    One ulp of a bf16 (fp16) number x is at most 2^-7 |x| (2^-10 |x|): the bound used here."""
    ref = util.f32(ref_bits, dt)
    assert np.isfinite(got).all()
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    rms = float(np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30))
    REPORT[f"full_vs_oracle/{tag}"] = dict(max_abs=err, max_o=scale, rel_rms=rms, elements=int(ref.size))
    assert err <= tol_rel * scale + (2 ** -7 if dt == 1 else 2 ** -10) * scale, f"{tag}: max|diff| {err:.3e} vs max|o| {scale:.3e}"


def test_config2_full_vs_oracle(oracle_mod):
    """BASELINE.json configs[1], every (batch, head): sageattn_qk_int8_pv_fp16_cuda against the oracle (B2 H32 N4096 D128 causal)."""
    q, k, v = rand_qkv(2, 32, 32, 4096, 4096, 128, 0, seed=2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=True, pv_accum_dtype="fp32", return_lse=True)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 0, is_causal=True, pv="f16",
                                                qk_quant_gran="per_thread", return_lse=True, km=km)
    _assert_vs_oracle("c2_f16pv_b2h32n4096d128_causal", o.float().cpu().numpy(), ref, 0)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3


@pytest.mark.parametrize("form", ["exact", "folded"])
def test_config3_full_vs_oracle(oracle_mod, form):
    """BASELINE.json configs[2] -- the headline configuration, all 64 (batch, head) units: FP8 PV, two-level
    accumulation, per-thread scales, bf16, B2 H32 N8192 D128 causal (the fused-Q default route).  exact: the default, against the exact
    oracle (the reference's formula); folded: the opt-in variant against the oracle mode that mirrors it."""
    q, k, v = rand_qkv(2, 32, 32, 8192, 8192, 128, 1, seed=3)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=True, pv_accum_dtype="fp32+fp32", return_lse=True, fp8_scores=form)
    if form == SCORES:
        o_default = sa.sageattn(qd, kd, vd, is_causal=True)
        torch.cuda.synchronize()
        assert torch.equal(o_default, o), "sageattn() dispatches to the FP8 two-level path in the default score form"
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 1, is_causal=True, pv="f8",
                                                qk_quant_gran="per_thread", return_lse=True, km=km, fp8_scores=form)
    _assert_vs_oracle(f"c3_f8pv_b2h32n8192d128_causal_{form}", o.float().cpu().numpy(), ref, 1)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3


def _varlen_km(k, cu_q, cu_k):
    """The K mean of a sageattn_varlen call as the call forms it: over all packed tokens, summed over the per-sequence slabs of its plan."""
    plan = sq.varlen_plan(cu_q, cu_k, total_q=int(cu_q[-1].item()), total_k=k.shape[0])
    return sq.channel_mean_packed(k, cu_k, plan)


@pytest.mark.parametrize("causal", [True, False])
def test_config4_full_vs_oracle(oracle_mod, causal):
    """BASELINE.json configs[3]: sageattn_varlen, Hq=32 Hkv=8 D=128 bf16, all eight sequences 256..16384 (one ragged).
    Causal: every head; non-causal (twice the oracle work): the first two GQA groups (8 query heads) of the full call."""
    lens = [256, 512, 1000, 1024, 2048, 4096, 8192, 16384]
    total = sum(lens)
    g = torch.Generator().manual_seed(4)
    q = torch.randn(total, 32, 128, generator=g).to(torch.bfloat16)
    k = (torch.randn(total, 8, 128, generator=g) + torch.randn(1, 8, 128, generator=g)).to(torch.bfloat16)
    v = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o = sa.sageattn_varlen(qd, kd, vd, cu.to(DEV), cu.to(DEV), max(lens), max(lens), is_causal=causal)
    torch.cuda.synchronize()
    km = util.bits(_varlen_km(kd, cu.to(DEV), cu.to(DEV)))          # [1, Hkv, D]: the mean over ALL packed tokens
    hq, hk = (32, 8) if causal else (8, 2)
    ref = oracle_mod.sageattn_varlen(util.bits(q[:, :hq]), util.bits(k[:, :hk]), util.bits(v[:, :hk]), 1, cu.numpy(), cu.numpy(),
                                     is_causal=causal, km=np.ascontiguousarray(km[:, :hk]))
    _assert_vs_oracle(f"c4_varlen_gqa_{'causal' if causal else 'noncausal'}", o[:, :hq].float().cpu().numpy(), ref, 1)


@pytest.mark.parametrize("form", ["exact", "folded"])
def test_config5_cogvideox_shape_vs_oracle(oracle_mod, form):
    """BASELINE.json configs[4]: the CogVideoX1.5-shaped drop-in call, B2 H48 N=17776 (= 277*64 + 48: a
    ragged last tile in both dimensions) D64 bf16 non-causal through sageattn(); the oracle checks eight heads of it.  exact: the default
    route against the exact oracle; folded: the opt-in variant against the oracle mode that mirrors it."""
    B, H, N, D = 2, 48, 17776, 64
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, H, N, D, generator=g).to(torch.bfloat16)
    k = (torch.randn(B, H, N, D, generator=g) + 2.0 * torch.randn(B, H, 1, D, generator=g)).to(torch.bfloat16)
    v = torch.randn(B, H, N, D, generator=g).to(torch.bfloat16)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o = sa.sageattn(qd, kd, vd, is_causal=False, **({} if form == SCORES else dict(fp8_scores=form)))
    torch.cuda.synchronize()
    assert o.shape == q.shape and o.dtype == torch.bfloat16 and torch.isfinite(o.float()).all()
    hs = [0, 7, 13, 22, 31, 38, 41, 47]
    b = 1
    km = util.bits(sq.channel_mean(kd))[b:b + 1, hs]
    ref, _, _ = oracle_mod.sageattn_dense(util.bits(q[b:b + 1, hs]), util.bits(k[b:b + 1, hs]), util.bits(v[b:b + 1, hs]), 1,
                                          is_causal=False, pv="f8", qk_quant_gran="per_thread", km=np.ascontiguousarray(km), fp8_scores=form)
    _assert_vs_oracle(f"c5_cogvideox_b2h48n17776d64_{form}", o[b:b + 1, hs].float().cpu().numpy(), ref, 1)
    truth = util.sdpa_f32(qd[b:b + 1, hs[:2]], kd[b:b + 1, hs[:2]], vd[b:b + 1, hs[:2]], False).cpu().numpy()
    got = o[b:b + 1, hs[:2]].float().cpu().numpy()
    cos, rel = util.cos_sim(got, truth), util.rmse(got, truth) / float(np.sqrt((truth ** 2).mean()))
    REPORT[f"full/c5_cogvideox_{form}"] = dict(cos=cos, rel_rmse=rel, rmse=util.rmse(got, truth))
    assert cos >= 0.999 and rel <= 0.05
    assert util.rmse(got, truth) <= 5e-3            # SURVEY 8c's absolute RMSE bar, as written


@pytest.mark.parametrize("name", ["varlenx_nc_d128_bf16", "varlenx_c_d64_f16"])
def test_varlen_cu_q_differs_from_cu_k(oracle_mod, name):
    """cu_seqlens_q != cu_seqlens_k (per-sequence Lq != Lk, causal top-left aligned) against the reference's Triton output
    (fixture generated by tests/golden/gen_golden.py) and against the oracle."""
    z, (nseq, Hq, Hkv, tq, tk, D, dt, causal) = util.golden(name)
    q, k, v = (util.from_bits(z[n], dt, DEV) for n in ("q", "k", "v"))
    cu_q, cu_k = torch.from_numpy(z["cu_q"]).to(DEV), torch.from_numpy(z["cu_k"]).to(DEV)
    mq, mk = int(np.diff(z["cu_q"]).max()), int(np.diff(z["cu_k"]).max())
    o = sa.sageattn_varlen(q, k, v, cu_q, cu_k, mq, mk, is_causal=bool(causal))
    torch.cuda.synchronize()
    got, ref = o.float().cpu().numpy(), util.f32(z["o"], dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"golden/{name}"] = dict(max_abs=err, max_o=scale)
    assert err <= 2e-3 * scale + (2 ** -7 * scale if dt == 1 else 0.0)
    km = util.from_bits(oracle_km_packed(z["k"], dt), dt, DEV)
    q8, qs, k8, ks, cu_qs, cu_ks = sq.per_block_int8_varlen(q, k, cu_q, cu_k, mq, mk, km=km, sm_scale=D ** -0.5)
    assert (cu_qs.cpu().numpy() == z["cu_qs"]).all() and (cu_ks.cpu().numpy() == z["cu_ks"]).all()
    assert (q8.cpu().numpy() == z["q_int8"]).all() and (qs.cpu().numpy() == z["q_scale"]).all()
    assert (k8.cpu().numpy() == z["k_int8"]).all() and (ks.cpu().numpy() == z["k_scale"]).all()
    ref_o = oracle_mod.sageattn_varlen(z["q"], z["k"], z["v"], dt, z["cu_q"], z["cu_k"], is_causal=bool(causal),
                                       km=util.bits(_varlen_km(k, cu_q, cu_k)))
    _assert_vs_oracle(f"varlen_cross/{name}", got, ref_o, dt)


# ------------------------------------------------------------------------------------------------ split-KV
def _split_oracle(O, q, k, v, dt, S, km, causal=False):
    """The split-KV algorithm restated with the oracle: quantise once (K mean, INT8 groups, per-channel FP8 V of the WHOLE
    tensors), run the oracle's attention per key-range chunk with fp16 partial outputs, merge by log-sum-exp in float64.
    Causal: chunk s holds keys s*Lc .., so only the query rows >= s*Lc see it, top-left aligned against the chunk."""
    _, _, aux = O.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8",
                                 qk_quant_gran="per_thread", km=km, fp8_scores=SCORES)
    B, Hq, Lq, D = q.shape
    Lk = k.shape[2]
    Lc = Lk // S
    parts = np.zeros((S, B, Hq, Lq, D), dtype=np.float64)
    lse = np.full((S, B, Hq, Lq), -np.inf, dtype=np.float64)
    for s in range(S):
        sl = slice(s * Lc, (s + 1) * Lc)
        r0 = s * Lc if causal else 0
        if r0 >= Lq:
            continue
        g0 = int(aux["gk"][s * Lc])
        gk = (aux["gk"][sl] - g0).astype(np.int32)
        ks = np.ascontiguousarray(aux["ks"][:, :, g0:g0 + int(gk.max()) + 1])
        o_s, lse_s = O.attn(np.ascontiguousarray(aux["q8"][:, :, r0:]), np.ascontiguousarray(aux["k8"][:, :, sl]),
                            np.ascontiguousarray(aux["v8"][:, :, sl]), aux["qs"], np.ascontiguousarray(aux["gq"][r0:]), ks, gk,
                            causal=causal, c=aux["c"], pv_mode=O.PV_F8_TWO_LEVEL, out_dtype=0, v_scale=aux["vs"], return_lse=True,
                            score_mode=O.SCORES_EXACT if SCORES == "exact" else O.SCORES_FOLDED)
        parts[s, :, :, r0:] = util.f32(o_s, 0)
        lse[s, :, :, r0:] = lse_s
    m = lse.max(axis=0)
    w = np.exp2(lse - m)
    o = (parts * w[..., None]).sum(axis=0) / w.sum(axis=0)[..., None]
    return o, m + np.log2(w.sum(axis=0))


@pytest.mark.parametrize("case", [(1, 4, 4, 128, 4096, 128, 1, 4, False), (2, 4, 2, 200, 2048, 64, 0, 8, False), (1, 2, 1, 64, 1024, 128, 1, 2, False),
                                  (1, 4, 2, 1024, 1024, 128, 1, 4, True), (2, 2, 2, 2048, 2048, 64, 0, 2, True), (1, 3, 3, 1000, 1024, 128, 1, 8, True),
                                  # chunks of an odd number of 64-key tiles: the chunk boundary cuts a 128-row q block in the middle, so waves 0 and 1
                                  # of the straddling block see a chunk in which every key is masked (m stays at its start value, lse = -inf)
                                  (1, 2, 2, 384, 384, 128, 0, 2, True), (1, 2, 1, 960, 960, 128, 1, 5, True), (2, 2, 2, 320, 320, 64, 0, 5, True)],
                         ids=["d128_bf16_s4", "gqa_d64_f16_s8", "gqa_d128_s2", "causal_gqa_d128_s4", "causal_d64_s2", "causal_lq1000_s8",
                              "causal_odd_tiles_384_s2", "causal_odd_tiles_960_s5", "causal_odd_tiles_d64_s5"])
def test_split_kv_vs_split_oracle_and_sdpa(oracle_mod, case):
    """Split-KV (chunks of the key range folded into the kv-head dimension + one log-sum-exp merge) against the same
    algorithm restated with the oracle (tolerance of the kernel tests), against the unsplit call and fp32 SDPA
    (the FP8 accuracy bounds: a different split changes which running maximum each P is rounded against).  Causal: the
    mask runs in global key coordinates, chunks behind the diagonal contribute nothing."""
    B, Hq, Hkv, Lq, Lk, D, dt, S, causal = case
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=500 + S, kbias=1.0)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32+fp32", return_lse=True, split_kv=S)
    o1, lse1 = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32+fp32", return_lse=True, split_kv=0)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, lse_ref = _split_oracle(oracle_mod, q, k, v, dt, S, km, causal)
    got = o.float().cpu().numpy()
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"split_kv/{B}x{Hq}x{Lq}x{Lk}_d{D}_s{S}{'_causal' if causal else ''}"] = dict(max_abs=err, max_o=scale)
    assert np.isfinite(got).all() and err <= 2e-3 * scale + (2 ** -7 if dt == 1 else 2 ** -10) * scale
    truth = util.sdpa_f32(qd, kd, vd, causal).cpu().numpy()
    for res in (got, o1.float().cpu().numpy()):
        assert util.cos_sim(res, truth) >= 0.999
    assert (lse - lse1).abs().max().item() <= 2e-2           # same quantity through two summation orders (+ fp8 noise on l)
    # ... and against the UNSPLIT oracle, i.e. the reference algorithm itself.  A split changes which running maximum every P is
    # rounded to e4m3 against, so the two differ by FP8 rounding noise of P (2^-4 relative per element, averaged over the row), not
    # by the 2e-3 of same-operand comparisons: rel-RMS up to 2.8e-2 measured, beyond the 1e-2 a DEFAULT FP8 route may differ from the exact
    # schedule by -- which is why FP8 split-KV is opt-in since round 5 (core._split_kv_plan).  Stated bound of the opt-in route (DESIGN.md 4,
    # divergence list): rel-RMS <= 4e-2, max <= 8e-2 * max|o| (the FP8 bound vs fp32 SDPA is 5e-2); measured values are in the parity report.
    ref_u, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8",
                                            qk_quant_gran="per_thread", km=km, fp8_scores=SCORES)
    ref_u = util.f32(ref_u, dt)
    d_u = got - ref_u
    rel_rms = float(np.sqrt((d_u ** 2).mean()) / np.sqrt((ref_u ** 2).mean()))
    REPORT[f"split_kv_vs_unsplit_oracle/{B}x{Hq}x{Lq}x{Lk}_d{D}_s{S}{'_causal' if causal else ''}"] = dict(
        rel_rms=rel_rms, max_abs=float(np.abs(d_u).max()), max_o=float(np.abs(ref_u).max()))
    assert rel_rms <= 4e-2 and np.abs(d_u).max() <= 8e-2 * np.abs(ref_u).max(), (rel_rms, float(np.abs(d_u).max()), float(np.abs(ref_u).max()))
    with pytest.raises(ValueError):
        sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, split_kv=7)
    if Lk % 128 == 64:       # an explicit split of a key range that is not a whole number of tiles per chunk is an error, not a silent no-op
        with pytest.raises(ValueError):
            sa.sageattn_qk_int8_pv_fp8_cuda(qd[:, :, :, :], kd[:, :, :Lk - 1], vd[:, :, :Lk - 1], split_kv=2)
    # the same call on [B, L, H, D] tensors: identical bits (the INT8 K and the V image are stored head-major either way)
    o_nhd = sa.sageattn_qk_int8_pv_fp8_cuda(qd.transpose(1, 2).contiguous(), kd.transpose(1, 2).contiguous(), vd.transpose(1, 2).contiguous(),
                                            tensor_layout="NHD", is_causal=causal, pv_accum_dtype="fp32+fp32", split_kv=S)
    assert torch.equal(o_nhd.transpose(1, 2), o)


@pytest.mark.parametrize("case", [(1, 4, 4, 128, 4096, 128, 0, 4, False), (1, 4, 2, 256, 2048, 64, 1, 8, False), (1, 2, 2, 1024, 1024, 128, 0, 4, True)],
                         ids=["d128_f16_s4", "gqa_d64_bf16_s8", "causal_d128_s4"])
def test_split_kv_fp16_pv_vs_unsplit_oracle(oracle_mod, case):
    """Split-KV of the FP16-PV entry point: P is rounded to fp16 (2^-11), so the split result meets the UNSPLIT oracle at the kernel
    tolerance widened only by the fp16 partial outputs of the chunks (one more rounding at 2^-11 of each chunk's |o|)."""
    B, Hq, Hkv, Lq, Lk, D, dt, S, causal = case
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=700 + S, kbias=1.0)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32", return_lse=True, split_kv=S)
    o1, lse1 = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32", return_lse=True, split_kv=0)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean(kd))
    ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16",
                                                qk_quant_gran="per_thread", return_lse=True, km=km)
    got, ref = o.float().cpu().numpy(), util.f32(ref, dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    REPORT[f"split_kv_fp16/{B}x{Hq}x{Lq}x{Lk}_d{D}_s{S}{'_causal' if causal else ''}"] = dict(max_abs=err, max_o=scale)
    assert np.isfinite(got).all() and err <= 4e-3 * scale + (2 ** -7 if dt == 1 else 2 ** -10) * scale
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= 5e-3
    assert (o.float() - o1.float()).abs().max().item() <= 4e-3 * scale + (2 ** -7 if dt == 1 else 2 ** -10) * scale


def test_split_kv_auto_plan_and_merge_kernel_edge_cases():
    """The planner (host logic) and the merge kernel alone: GQA chunk layout, a tail chunk, rows a chunk did not see (-inf)."""
    from sageattention_amd import core, _cabi
    assert core._split_kv_plan(1, 32, 128, 32768, False, None) == 16          # 32 workgroups, 512 tiles -> 16 chunks of 32 tiles
    assert core._split_kv_plan(2, 32, 8192, 8192, False, None) == 0           # the grid already fills the chip
    assert core._split_kv_plan(1, 32, 128, 32768, True, None) == 0            # causal with Lq != Lk is not split
    assert core._split_kv_plan(1, 8, 8192, 8192, True, None) == 0             # causal: only on request (measured slower)
    assert core._split_kv_plan(1, 8, 8192, 8192, True, 4) == 4
    assert core._split_kv_plan(1, 8, 128, 32768 + 32, False, None) == 0       # ragged key range
    B, Hkv, group, S, L, D = 2, 2, 3, 4, 37, 64
    H = Hkv * group
    g = torch.Generator().manual_seed(9)
    o_part = torch.randn(B, Hkv, S, group, L, D, generator=g).to(torch.float16).to(DEV)
    lse_part = (3.0 * torch.randn(B, Hkv, S, group, L, generator=g)).to(DEV)
    lse_part[0, 0, 1] = float("-inf")
    lse_part[1, 1, :, 2, 5] = float("-inf")                                   # a row no chunk saw
    o_tail = torch.randn(B, H, L, D, generator=g).to(torch.float16).to(DEV)
    lse_tail = torch.randn(B, H, L, generator=g).to(DEV)
    for with_tail in (False, True):
        o = torch.empty(B, H, L, D, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B, H, L, dtype=torch.float32, device=DEV)
        rc = _cabi.load().sage_merge_split(o_part.data_ptr(), lse_part.data_ptr(), o_tail.data_ptr() if with_tail else None,
                                           lse_tail.data_ptr() if with_tail else None, o.data_ptr(), lse.data_ptr(), B, S, H, group, L, D,
                                           o.stride(0), o.stride(1), o.stride(2), _cabi.DTYPE_BF16, torch.cuda.current_stream().cuda_stream)
        _cabi.check(rc, "sage_merge_split")
        torch.cuda.synchronize()
        op = o_part.double().permute(2, 0, 1, 3, 4, 5).reshape(S, B, H, L, D)
        lp = lse_part.double().permute(2, 0, 1, 3, 4).reshape(S, B, H, L)
        if with_tail:
            op, lp = torch.cat([op, o_tail.double()[None]]), torch.cat([lp, lse_tail.double()[None]])
        m = lp.max(dim=0).values
        w = torch.exp2(lp - m.clamp_min(-1e30))
        w = torch.where(torch.isinf(lp), torch.zeros_like(w), w)
        ws = w.sum(dim=0)
        want = torch.where(ws[..., None] > 0, (op * w[..., None]).sum(dim=0) / ws[..., None].clamp_min(1e-300), torch.zeros_like(op[0]))
        assert (o.double() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item() + 1e-6
        want_lse = torch.where(ws > 0, m + torch.log2(ws.clamp_min(1e-300)), torch.full_like(m, float("-inf")))
        fin = torch.isfinite(want_lse)
        assert torch.equal(torch.isfinite(lse.double()), fin) and (lse.double()[fin] - want_lse[fin]).abs().max().item() <= 1e-4


# ------------------------------------------------------------------------------------------------ LSE merge / ring caller
@pytest.mark.parametrize("dt,layout,D,L", [(0, "HND", 128, 333), (1, "NHD", 64, 130), (1, "HND", 96, 17)])
def test_merge_states_matches_formula(dt, layout, D, L):
    """sage_merge_states (sequence-parallel combine of return_lse results) vs the fp32 formula, incl. -inf rows."""
    from sageattention_amd import ring
    g = torch.Generator().manual_seed(12)
    B, H = 2, 3
    oa, ob = torch.randn(B, H, L, D, generator=g).to(T(dt)), torch.randn(B, H, L, D, generator=g).to(T(dt))
    la, lb = 4.0 * torch.randn(B, H, L, generator=g), 4.0 * torch.randn(B, H, L, generator=g)
    lb[0, 0, :5] = float("-inf")                      # shard contributed nothing to these rows
    la[1, 2, 3] = float("-inf")
    la[1, 1, 7] = lb[1, 1, 7] = float("-inf")          # nobody did
    acc_ref, lse_ref = torch.empty(B, H, L, D), torch.empty(B, H, L)
    util.merge_states_torch(acc_ref, lse_ref, oa if layout == "HND" else oa.transpose(1, 2), la, layout, first=True)
    out_ref = torch.empty_like(oa if layout == "HND" else oa.transpose(1, 2).contiguous())
    util.merge_states_torch(acc_ref, lse_ref, ob if layout == "HND" else ob.transpose(1, 2), lb, layout, out=out_ref)

    acc = torch.empty(B, H, L, D, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, H, L, dtype=torch.float32, device=DEV)
    oad, obd = to_dev(oa, layout), to_dev(ob, layout)
    out = torch.empty_like(oad)
    ring.merge_states(acc, lse, oad, la.to(DEV), layout, first=True)
    assert torch.equal(acc.cpu(), oa.float()) and torch.equal(lse.cpu(), la)
    ring.merge_states(acc, lse, obd, lb.to(DEV), layout, out=out)
    torch.cuda.synchronize()
    assert torch.isfinite(acc).all()
    assert (acc.cpu() - acc_ref).abs().max().item() <= 1e-5 * acc_ref.abs().max().item()
    fin = torch.isfinite(lse_ref)
    assert torch.equal(torch.isfinite(lse.cpu()), fin) and (lse.cpu()[fin] - lse_ref[fin]).abs().max().item() <= 1e-5
    ulp = 2.0 ** (-10 if dt == 0 else -7)
    assert ((out.float().cpu() - out_ref.float()).abs() <= ulp * out_ref.float().abs().clamp_min(2.0 ** -14)).all()


@pytest.mark.parametrize("causal", [False, True])
def test_ring_steps_on_one_gpu_match_full_attention(causal):
    """The ring caller's per-rank computation replayed on one device: rank r of 3 attends shard by shard with
    sageattn(return_lse=True) and merges with the HIP kernel; the result must meet the FP8 accuracy bound against
    fp32 SDPA over the whole sequence (shard-wise K smoothing / scales differ from the unsharded call, so the
    comparison is against the truth, not bit-for-bit against a single call)."""
    from sageattention_amd import ring
    W, B, H, Lc, D = 3, 1, 4, 384, 128
    q, k, v = rand_qkv(B, H, H, W * Lc, W * Lc, D, 1, seed=21, kbias=1.0)
    truth = util.sdpa_f32(q, k, v, causal).numpy()
    for r in range(W):
        qs = q[:, :, r * Lc:(r + 1) * Lc].to(DEV)
        acc = torch.empty(B, H, Lc, D, dtype=torch.float32, device=DEV)
        lse = torch.empty(B, H, Lc, dtype=torch.float32, device=DEV)
        out = torch.empty_like(qs)
        sched = [(s, j, m) for s, j, m in ring.shard_schedule(r, W, causal) if m != "skip"]
        for idx, (s, j, mode) in enumerate(sched):
            ks, vs = k[:, :, j * Lc:(j + 1) * Lc].to(DEV), v[:, :, j * Lc:(j + 1) * Lc].to(DEV)
            o_s, lse_s = sa.sageattn(qs, ks, vs, is_causal=(mode == "causal"), return_lse=True)
            ring.merge_states(acc, lse, o_s, lse_s, first=(idx == 0), out=out if idx == len(sched) - 1 else None)
        got = out.float().cpu().numpy()
        want = truth[:, :, r * Lc:(r + 1) * Lc]
        rel = util.rmse(got, want) / float(np.sqrt((want ** 2).mean()))
        REPORT[f"ring/{'c' if causal else 'nc'}/rank{r}"] = dict(rel_rmse=rel, cos=util.cos_sim(got, want))
        assert util.cos_sim(got, want) >= 0.999 and rel <= 0.05


# ------------------------------------------------------------------------------------------------ randomized sweep + graphs
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SAGE_RANDOM_SEEDS", "100")))))      # (SAGE_RANDOM_SEEDS=400: a one-off stress run)
def test_random_shapes_vs_oracle(oracle_mod, seed):
    """Seeded random problems (shapes around the 64/128 tile edges and the steady/general loop boundary, GQA, both head
    dims, every granularity and accumulation mode, both layouts) against the oracle on identical operands."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.choice([64, 128]))
    Hkv = int(rng.integers(1, 4))
    Hq = Hkv * int(rng.choice([1, 2, 4]))
    B = int(rng.integers(1, 3))
    Lk = int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(190, 330)), int(rng.integers(500, 900))]))
    causal = bool(rng.integers(0, 2))
    Lq = Lk if (causal and rng.random() < 0.7) else int(rng.integers(1, 420))
    dt = int(rng.integers(0, 2))
    gran = str(rng.choice(["per_block", "per_warp", "per_thread"]))
    pv = str(rng.choice(["f8_two", "f8_single", "f8f_two", "f8f_single", "f16_two", "f16_single"]))
    layout = str(rng.choice(["HND", "NHD"]))
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=seed, kbias=float(rng.random() * 2))
    fp8 = pv.startswith("f8")
    km = util.bits(sq.channel_mean(k.to(DEV)))
    scores = _scores_of(pv)
    o_bits, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal,
                                                   pv="f8" if fp8 else "f16", qk_quant_gran=gran, return_lse=True, km=km,
                                                   warpq=16 if (pv == "f16_two" and D == 128) else 32, fp8_scores=scores or "exact",
                                                   single_level=fp8 and pv.endswith("single"))
    fn = sa.sageattn_qk_int8_pv_fp8_cuda if fp8 else sa.sageattn_qk_int8_pv_fp16_cuda
    accum = PV_ACCUM[pv]
    kw = dict(fp8_scores=scores) if fp8 else {}
    o, lse = fn(to_dev(q, layout), to_dev(k, layout), to_dev(v, layout), tensor_layout=layout, is_causal=causal,
                qk_quant_gran=gran, pv_accum_dtype=accum, return_lse=True, **kw)
    torch.cuda.synchronize()
    got, ref = to_hnd(o, layout).float().cpu().numpy(), util.f32(o_bits, dt)
    desc = f"B{B} Hq{Hq} Hkv{Hkv} Lq{Lq} Lk{Lk} D{D} dt{dt} causal{causal} {gran} {pv} {layout}"
    assert np.isfinite(got).all(), desc
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    assert err <= 2e-3 * scale + util.out_ulp(scale, dt), f"{desc}: {err:.3e} vs {scale:.3e}"
    # (the q.km^T correction of the LSE is rounded to the input dtype: bf16 keeps 8 bits of a term of a few units -- seed 96 of a 400-seed run: 5.5e-3)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= (5e-3 if dt == 0 else 2e-2), desc


@pytest.mark.parametrize("L", [576, 622, 739, 1100])
@pytest.mark.parametrize("pv,D", [("f16_two", 128), ("f16_single", 128), ("f16_single", 64), ("f8_two", 128), ("f8_two", 64)])
def test_causal_ragged_last_block_with_an_odd_count_of_pipelined_tiles(oracle_mod, pv, D, L):
    """Causal, Lq = Lk not a multiple of 128: only the last query block runs an ODD number of tiles through the software-pipelined loop (7, 7, 9
    and 15 here; whole blocks always run an even number), which takes the loop's one-tile prologue and its register rename.  The rename's
    copies of MFMA results were once scheduled by the compiler above the wait states in front of them (FP16 PV, D = 128, causal: errors of
    1.4 % of max|o| in the last block's rows, found by seeds 94 and 174 of a 400-seed run of the test above; tools/mfma_hazard_lint.py)."""
    dt = L & 1
    q, k, v = rand_qkv(1, 2, 1, L, L, D, dt, seed=L, kbias=1.0)
    fp8 = pv.startswith("f8")
    km = util.bits(sq.channel_mean(k.to(DEV)))
    for gran in ("per_thread", "per_warp"):
        o_bits, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=True, pv="f8" if fp8 else "f16",
                                                       qk_quant_gran=gran, return_lse=True, km=km,
                                                       warpq=16 if (pv == "f16_two" and D == 128) else 32, fp8_scores=SCORES)
        fn = sa.sageattn_qk_int8_pv_fp8_cuda if fp8 else sa.sageattn_qk_int8_pv_fp16_cuda
        o, lse = fn(q.to(DEV), k.to(DEV), v.to(DEV), is_causal=True, qk_quant_gran=gran, pv_accum_dtype=PV_ACCUM[pv], return_lse=True)
        torch.cuda.synchronize()
        got, ref = o.float().cpu().numpy(), util.f32(o_bits, dt)
        scale = float(np.abs(ref).max())
        bar = 2e-3 * scale + util.out_ulp(scale, dt)
        last = (L - 1) // 128 * 128
        over = np.abs(got - ref) > bar
        assert not over.any(), f"{pv} D{D} L{L} {gran}: {int(over.sum())} elements over the bar, {int(over[:, :, last:].sum())} of them in the last block"
        assert np.abs(lse.cpu().numpy() - lse_ref).max() <= (5e-3 if dt == 0 else 2e-2)


@pytest.mark.parametrize("ns", list(range(1, 15)))
@pytest.mark.parametrize("pv,D", [("f16_two", 128), ("f16_single", 128), ("f16_single", 64), ("f8_two", 128), ("f8_two", 64)])
def test_every_count_of_pipelined_tiles(oracle_mod, pv, D, ns):
    """Non-causal, Lk = 64 (ns + 2): exactly `ns` tiles run through the software-pipelined loop -- since round 6 six bodies per trip (ring slot
    and register set compile-time constants in each) and a remainder loop of up to five single bodies, each renamed behind it on the matrix
    pipe (FP16 PV: behind a peeled first body).  ns = 1 ... 14 takes every remainder behind zero, one and two trips, in every loop form the
    library instantiates (D = 64 FP16 PV keeps the two-body form)."""
    Lk, Lq, dt = 64 * (ns + 2), 200, ns & 1
    q, k, v = rand_qkv(1, 2, 2, Lq, Lk, D, dt, seed=1000 + ns, kbias=1.0)
    fp8 = pv.startswith("f8")
    km = util.bits(sq.channel_mean(k.to(DEV)))
    o_bits, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=False, pv="f8" if fp8 else "f16",
                                                   qk_quant_gran="per_thread", return_lse=True, km=km,
                                                   warpq=16 if (pv == "f16_two" and D == 128) else 32, fp8_scores=SCORES)
    fn = sa.sageattn_qk_int8_pv_fp8_cuda if fp8 else sa.sageattn_qk_int8_pv_fp16_cuda
    o, lse = fn(q.to(DEV), k.to(DEV), v.to(DEV), is_causal=False, qk_quant_gran="per_thread", pv_accum_dtype=PV_ACCUM[pv], return_lse=True)
    torch.cuda.synchronize()
    got, ref = o.float().cpu().numpy(), util.f32(o_bits, dt)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    assert np.isfinite(got).all()
    assert err <= 2e-3 * scale + util.out_ulp(scale, dt), f"{pv} D{D} ns{ns}: {err:.3e} vs {scale:.3e}"
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= (5e-3 if dt == 0 else 2e-2)


@pytest.mark.parametrize("Lk", [128, 129, 160, 191, 192, 193, 200, 255, 256, 257, 320, 383, 449, 641, 704, 705])
@pytest.mark.parametrize("D", [128, 64])
def test_last_tiles_of_a_non_causal_fp8_call_through_the_pipelined_body(oracle_mod, D, Lk):
    """Non-causal FP8 PV: the two whole tiles the steady loop leaves behind (it looks two tiles ahead) and a ragged last tile run the pipelined
    body too since round 6 -- keys past Lk masked in front of the row maximum, the ragged tile requested with its rows clamped to the last key
    (a negative per-lane offset there was a memory fault once: the VGPR offset of the SGPR-base LDS-DMA is unsigned).  Lk from two tiles up:
    no steady tile at all, every remainder, ragged tails of 1 ... 63 keys; K is the last tensor allocated, so rows past its end are not ours."""
    Lq, dt = 136, Lk & 1
    q, k, v = rand_qkv(1, 2, 2, Lq, Lk, D, dt, seed=2000 + Lk, kbias=1.0)
    km = util.bits(sq.channel_mean(k.to(DEV)))
    for gran in ("per_thread", "per_warp"):
        o_bits, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=False, pv="f8", qk_quant_gran=gran,
                                                       return_lse=True, km=km, fp8_scores=SCORES)
        qd, vd = q.to(DEV), v.to(DEV)
        kd = k.to(DEV)
        o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=False, qk_quant_gran=gran, pv_accum_dtype="fp32+fp32", return_lse=True)
        torch.cuda.synchronize()
        got, ref = o.float().cpu().numpy(), util.f32(o_bits, dt)
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        assert np.isfinite(got).all()
        assert err <= 2e-3 * scale + util.out_ulp(scale, dt), f"D{D} Lk{Lk} {gran}: {err:.3e} vs {scale:.3e}"
        assert np.abs(lse.cpu().numpy() - lse_ref).max() <= (5e-3 if dt == 0 else 2e-2)


@pytest.mark.parametrize("L", [128, 192, 256, 320, 384, 512, 896])
@pytest.mark.parametrize("pv,D", [("f16_two", 128), ("f16_single", 128), ("f8_two", 128), ("f8_two", 64)])
def test_diagonal_tiles_of_a_causal_call_through_the_pipelined_body(oracle_mod, pv, D, L):
    """Causal, whole 64-key tiles: a work item's last two tiles (the diagonal ones when Lq = Lk) take the pipelined body since round 6 -- scores
    behind the diagonal replaced by a large negative pattern in front of the row maximum -- FP8 PV also in the first query block, which has
    no steady tile in front of them.  L = 192 / 320: the last query block is half a block (general tiles there); Lq != Lk: the last two
    tiles are not the diagonal ones for every block."""
    dt = (L >> 6) & 1
    fp8 = pv.startswith("f8")
    for Lq, Lk in ((L, L), (L, L + 128), (L + 64, L)):
        q, k, v = rand_qkv(1, 2, 1, Lq, Lk, D, dt, seed=3000 + L + Lq, kbias=1.0)
        km = util.bits(sq.channel_mean(k.to(DEV)))
        o_bits, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=True, pv="f8" if fp8 else "f16",
                                                       qk_quant_gran="per_thread", return_lse=True, km=km,
                                                       warpq=16 if (pv == "f16_two" and D == 128) else 32, fp8_scores=SCORES)
        fn = sa.sageattn_qk_int8_pv_fp8_cuda if fp8 else sa.sageattn_qk_int8_pv_fp16_cuda
        o, lse = fn(q.to(DEV), k.to(DEV), v.to(DEV), is_causal=True, qk_quant_gran="per_thread", pv_accum_dtype=PV_ACCUM[pv], return_lse=True)
        torch.cuda.synchronize()
        got, ref = o.float().cpu().numpy(), util.f32(o_bits, dt)
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        assert np.isfinite(got).all()
        assert err <= 2e-3 * scale + util.out_ulp(scale, dt), f"{pv} D{D} Lq{Lq} Lk{Lk}: {err:.3e} vs {scale:.3e}"
        assert np.abs(lse.cpu().numpy() - lse_ref).max() <= (5e-3 if dt == 0 else 2e-2)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SAGE_RANDOM_SEEDS", "100")))))
def test_random_calls_of_the_other_entry_points_vs_oracle(oracle_mod, seed):
    """The seeded sweep above for the entry points it does not reach: the Triton-named API (per-block scales, Q quantised in the kernel),
    sageattn_varlen (packed sequences of random lengths incl. 1-token and empty-query ones, cu_q != cu_k when not causal), the sm90 entry
    point (its own scale groups) and attn_mask (bool / additive, broadcast shapes) -- each against the oracle on identical operands with the same K mean."""
    rng = np.random.default_rng(5000 + seed)
    kind = ("triton", "varlen", "sm90", "mask", "pad")[seed % 5]
    D = int(rng.choice([64, 128])) if kind != "pad" else int(rng.choice([8, 40, 56, 72, 80, 96, 100, 120]))    # (pad: sageattn() with other head sizes)
    Hkv = int(rng.integers(1, 4))
    Hq = Hkv * int(rng.choice([1, 2, 4]))
    dt = int(rng.integers(0, 2))
    causal = bool(rng.integers(0, 2))
    pick_len = lambda: int(rng.choice([int(rng.integers(1, 200)), int(rng.integers(190, 330)), int(rng.integers(500, 1200))]))
    if kind == "varlen":
        nseq = int(rng.integers(1, 6))
        lk = [pick_len() for _ in range(nseq)]
        lq = list(lk) if causal else [int(rng.choice([0, 1, pick_len()], p=[0.1, 0.1, 0.8])) for _ in range(nseq)]
        if sum(lq) == 0:
            lq[0] = 3
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(sum(lq), Hq, D, generator=g).to(T(dt))
        k = (torch.randn(sum(lk), Hkv, D, generator=g) + float(rng.random() * 2) * torch.randn(1, Hkv, D, generator=g)).to(T(dt))
        v = torch.randn(sum(lk), Hkv, D, generator=g).to(T(dt))
        cu_q = torch.tensor([0] + list(np.cumsum(lq)), dtype=torch.int32)
        cu_k = torch.tensor([0] + list(np.cumsum(lk)), dtype=torch.int32)
        desc = f"varlen lq{lq} lk{lk} Hq{Hq} Hkv{Hkv} D{D} dt{dt} causal{causal}"
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        o = sa.sageattn_varlen(qd, kd, vd, cu_q.to(DEV), cu_k.to(DEV), max(lq), max(lk), is_causal=causal)
        torch.cuda.synchronize()
        km = util.bits(_varlen_km(kd, cu_q.to(DEV), cu_k.to(DEV)))
        ref = oracle_mod.sageattn_varlen(util.bits(q), util.bits(k), util.bits(v), dt, cu_q.numpy(), cu_k.numpy(), is_causal=causal, km=km)
        _assert_vs_oracle(f"random/{seed}/{desc}", o.float().cpu().numpy(), ref, dt)
        return
    B = int(rng.integers(1, 3))
    Lk = pick_len()
    if kind == "mask":
        causal = False                                   # (core.py:310: no mask under is_causal)
    Lq = Lk if (causal and (kind == "triton" or rng.random() < 0.7)) else int(rng.integers(1, 420))
    layout = str(rng.choice(["HND", "NHD"]))
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=seed, kbias=float(rng.random() * 2))
    Dp = D if kind != "pad" else (64 if D <= 64 else 128)          # (the K mean of a padded call is taken over the padded tensor: zeros in the pad)
    km = util.bits(sq.channel_mean(torch.nn.functional.pad(k, (0, Dp - D)).to(DEV)))
    desc = f"{kind} B{B} Hq{Hq} Hkv{Hkv} Lq{Lq} Lk{Lk} D{D} dt{dt} causal{causal} {layout}"
    if kind == "mask":
        # bool (with all-False 128 x 64 tiles, which the kernel skips, and fully masked rows) or additive in q's dtype; broadcast over batch / heads at random
        mshape = (B if rng.random() < 0.5 else 1, Hq if rng.random() < 0.5 else 1, Lq, Lk)
        g = torch.Generator().manual_seed(seed + 9)
        if rng.random() < 0.5:
            m = torch.rand(mshape, generator=g) < 0.7
            m[..., : min(Lq, 130), 64:192] = False
            m[..., Lq // 2, :] = False
            kw = dict(mask_bool=m.numpy())
        else:
            m = (2.0 * torch.randn(mshape, generator=g)).to(T(dt))
            kw = dict(mask_add=util.f32(util.bits(m), dt))
        desc += f" mask {tuple(mshape)} {m.dtype}"
        ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, pv="f16_triton", qk_quant_gran="per_block",
                                                    return_lse=True, km=km, **kw)
        o, lse = sa.sageattn_qk_int8_pv_fp16_triton(to_dev(q, layout), to_dev(k, layout), to_dev(v, layout), tensor_layout=layout,
                                                    attn_mask=m.to(DEV), return_lse=True)
        torch.cuda.synchronize()
        _assert_vs_oracle(f"random/{seed}/{desc}", to_hnd(o, layout).float().cpu().numpy(), ref, dt)
        # (fully masked rows carry an LSE of about -1e6 / log2 e, where one fp32 ulp is 0.0625)
        lse, fin = lse.cpu().numpy(), np.isfinite(lse_ref)
        assert np.array_equal(np.isneginf(lse), ~fin), desc              # (a query block with every tile skipped: zeros, lse -inf)
        assert (np.abs(lse[fin] - lse_ref[fin]) <= (5e-3 if dt == 0 else 2e-2) + 2e-7 * np.abs(lse_ref[fin])).all(), desc
        return
    if kind == "pad":
        # sageattn(): head sizes padded to 64 / 128 (core.py:253-268 -> FP8 two-level, per-thread on this device), softmax scale of the ORIGINAL size
        # unless given, K smoothing optional
        smooth_k = bool(rng.random() < 0.7)
        sm_scale = None if rng.random() < 0.6 else float(0.05 + 0.2 * rng.random())
        desc += f" smooth_k{smooth_k} sm_scale{sm_scale}"
        ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8", qk_quant_gran="per_thread",
                                                    return_lse=True, km=km if smooth_k else None, smooth_k=smooth_k, sm_scale=sm_scale, fp8_scores=SCORES)
        if smooth_k:
            o, lse = sa.sageattn(to_dev(q, layout), to_dev(k, layout), to_dev(v, layout), tensor_layout=layout, is_causal=causal, sm_scale=sm_scale,
                                 return_lse=True)
        else:
            o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(to_dev(q, layout), to_dev(k, layout), to_dev(v, layout), tensor_layout=layout, is_causal=causal,
                                                     sm_scale=sm_scale, smooth_k=False, pv_accum_dtype="fp32+fp32", return_lse=True)
        assert o.shape[-1] == D
    elif kind == "triton":
        ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16_triton",
                                                    qk_quant_gran="per_block", return_lse=True, km=km)
        o, lse = sa.sageattn_qk_int8_pv_fp16_triton(to_dev(q, layout), to_dev(k, layout), to_dev(v, layout), tensor_layout=layout,
                                                    is_causal=causal, return_lse=True)
    else:
        gran = str(rng.choice(["per_warp", "per_thread"]))
        desc += " " + gran
        ref, lse_ref, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8", qk_quant_gran=gran,
                                                    return_lse=True, km=km, warpq=16, blkk=128, fp8_scores=SCORES)
        o, lse = sa.sageattn_qk_int8_pv_fp8_cuda_sm90(to_dev(q, layout), to_dev(k, layout), to_dev(v, layout), tensor_layout=layout,
                                                      is_causal=causal, qk_quant_gran=gran, pv_accum_dtype="fp32+fp32", return_lse=True)
    torch.cuda.synchronize()
    _assert_vs_oracle(f"random/{seed}/{desc}", to_hnd(o, layout).float().cpu().numpy(), ref, dt)
    assert np.abs(lse.cpu().numpy() - lse_ref).max() <= (5e-3 if dt == 0 else 2e-2), desc


def test_sageattn_is_hip_graph_capturable():
    """The dense pipeline (K mean, quantisers, V pre-pass, attention) has no host synchronisation and launches on the
    caller's stream, so a serving loop can capture it in a HIP graph and replay it with new inputs in place."""
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 8, 8, 1024, 1024, 128, 1, seed=5))
    sa.sageattn(q, k, v, is_causal=True)                       # warm-up outside capture (library load, allocator)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        sa.sageattn(q, k, v, is_causal=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        o_graph = sa.sageattn(q, k, v, is_causal=True)
    q2, k2, v2 = (t.to(DEV) for t in rand_qkv(1, 8, 8, 1024, 1024, 128, 1, seed=6))
    q.copy_(q2); k.copy_(k2); v.copy_(v2)
    g.replay()
    torch.cuda.synchronize()
    want = sa.sageattn(q2, k2, v2, is_causal=True)
    assert torch.equal(o_graph, want)


# ------------------------------------------------------------------------------------------------ fused Q quantisation
@pytest.mark.parametrize("shape", [(2, 4, 2, 300, 300, 128), (1, 3, 3, 129, 1000, 64), (1, 8, 8, 1024, 1024, 128), (2, 2, 1, 5, 70, 128)],
                         ids=["gqa300", "cross_d64", "n1024", "tiny"])
@pytest.mark.parametrize("dt,layout,causal", [(0, "HND", False), (1, "NHD", True), (1, "HND", True)])
def test_fused_q_quant_is_bit_identical_to_the_separate_quantiser(shape, dt, layout, causal):
    """sage_attn_fused_q_pv_f8 quantises Q in the kernel prologue with the arithmetic of the stand-alone per-thread
    quantiser: outputs and LSE must equal the two-kernel route bit for bit (so every oracle-parity statement about
    sageattn() carries over to the fused default route)."""
    B, Hq, Hkv, Lq, Lk, D = shape
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=77, kbias=1.0)
    qd, kd, vd = to_dev(q, layout), to_dev(k, layout), to_dev(v, layout)
    o1, l1 = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, tensor_layout=layout, is_causal=causal, pv_accum_dtype="fp32+fp32", return_lse=True)
    o0, l0 = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, tensor_layout=layout, is_causal=causal, pv_accum_dtype="fp32+fp32", return_lse=True,
                                             fuse_q_quant=False)
    torch.cuda.synchronize()
    assert torch.equal(o1, o0) and torch.equal(l1, l0)
    # without K smoothing, and with the fused v_mean epilogue (smooth_v needs the single-level kernel -> unfused route only)
    o3 = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, tensor_layout=layout, is_causal=causal, pv_accum_dtype="fp32+fp16", smooth_k=False)
    o4 = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, tensor_layout=layout, is_causal=causal, pv_accum_dtype="fp32+fp16", smooth_k=False,
                                         fuse_q_quant=False)
    assert torch.equal(o3, o4)
    # a strided q view (fused QKV projection output) goes through the fused kernel without a copy
    if layout == "NHD":
        qkv = torch.stack([qd, qd, qd], dim=2)                       # [B, L, 3, H, D]
        o2 = sa.sageattn_qk_int8_pv_fp8_cuda(qkv[:, :, 1], kd, vd, tensor_layout=layout, is_causal=causal, pv_accum_dtype="fp32+fp32")
        assert torch.equal(o2, o0)


@pytest.mark.parametrize("shape", [(2, 4, 2, 300, 300, 128), (1, 3, 3, 129, 1000, 64), (1, 8, 8, 1024, 1024, 128), (2, 2, 1, 5, 70, 128),
                                   (1, 2, 2, 640, 640, 64)], ids=["gqa300", "cross_d64", "n1024", "tiny", "d64_640"])
@pytest.mark.parametrize("dt,layout,causal", [(0, "HND", False), (1, "NHD", True), (1, "HND", True)])
def test_fused_per_block_q_quant_is_bit_identical_to_the_separate_quantiser(shape, dt, layout, causal):
    """sage_attn_fused_qblock_pv_f16 (the default route of the Triton-named API) quantises Q per 128-row block in the kernel prologue with
    the arithmetic of the stand-alone per-block quantiser (sm_scale * log2 e folded in first): outputs and LSE equal the two-kernel
    route bit for bit, so the reference-fixture parity of that route carries over."""
    B, Hq, Hkv, Lq, Lk, D = shape
    if causal and Lq != Lk:
        pytest.skip("the Triton-named API's causal mask is for self-attention")
    q, k, v = rand_qkv(B, Hq, Hkv, Lq, Lk, D, dt, seed=78, kbias=1.0)
    qd, kd, vd = to_dev(q, layout), to_dev(k, layout), to_dev(v, layout)
    o1, l1 = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, tensor_layout=layout, is_causal=causal, return_lse=True)
    o0, l0 = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, tensor_layout=layout, is_causal=causal, return_lse=True, fuse_q_quant=False)
    torch.cuda.synchronize()
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(o1, o0) and torch.equal(l1, l0)
    o3 = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, tensor_layout=layout, is_causal=causal, smooth_k=False, sm_scale=0.2)
    o4 = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, tensor_layout=layout, is_causal=causal, smooth_k=False, sm_scale=0.2, fuse_q_quant=False)
    assert torch.equal(o3, o4)
    # an all-zero query block (scale 0: every q_int8 is 0, as the stand-alone quantiser gives) and a strided q view
    qz = qd.clone()
    if layout == "HND":
        qz[:, :, :128] = 0
    else:
        qz[:, :128] = 0
    assert torch.equal(sa.sageattn_qk_int8_pv_fp16_triton(qz, kd, vd, tensor_layout=layout, is_causal=causal),
                       sa.sageattn_qk_int8_pv_fp16_triton(qz, kd, vd, tensor_layout=layout, is_causal=causal, fuse_q_quant=False))
    if layout == "NHD":
        qkv = torch.stack([qd, qd, qd], dim=2)                       # [B, L, 3, H, D]
        assert torch.equal(sa.sageattn_qk_int8_pv_fp16_triton(qkv[:, :, 1], kd, vd, tensor_layout=layout, is_causal=causal), o0)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dt,D,hq,hkv", [(1, 128, 8, 2), (0, 64, 4, 4)])
def test_varlen_fused_per_block_q_quant_is_bit_identical(causal, dt, D, hq, hkv):
    """sageattn_varlen's default route (sage_attn_fused_qblock_pv_f16_varlen: Q quantised per block in the attention kernel, no host sync)
    against the reference-shaped two-kernel route (per_block_int8_varlen's Q half + sage_attn_qk_int8_pv_f16_varlen): same bits."""
    lens = [1, 127, 128, 129, 700, 64, 1000]
    total = sum(lens)
    g = torch.Generator().manual_seed(123)
    T = torch.float16 if dt == 0 else torch.bfloat16
    q = torch.randn(total, hq, D, generator=g).to(T).to(DEV)
    k = (torch.randn(total, hkv, D, generator=g) + torch.randn(1, hkv, D, generator=g)).to(T).to(DEV)
    v = torch.randn(total, hkv, D, generator=g).to(T).to(DEV)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    o1 = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
    o0 = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal, fuse_q_quant=False)
    torch.cuda.synchronize()
    assert torch.isfinite(o1.float()).all() and torch.equal(o1, o0)
    if not causal:                     # cross attention: other key lengths, a strided q (slice of a fused QKV projection)
        klens = [5, 64, 200, 77, 1000, 640, 3]
        cuk = torch.tensor([0] + list(np.cumsum(klens)), dtype=torch.int32, device=DEV)
        k2, v2 = k[:sum(klens)].contiguous(), v[:sum(klens)].contiguous()
        qkv = torch.stack([q, q], dim=1)                              # [T, 2, H, D]
        o3 = sa.sageattn_varlen(qkv[:, 1], k2, v2, cu, cuk, max(lens), max(klens))
        o2 = sa.sageattn_varlen(q, k2, v2, cu, cuk, max(lens), max(klens), fuse_q_quant=False)
        assert torch.equal(o3, o2)


def test_varlen_plan_matches_the_torch_prefix_sums():
    """sage_varlen_plan (one launch) against the reference's torch ops (quant_per_block_varlen.py:68-73), a sort by length, and -- the work
    list, its header and the slab map -- against the same functions run on the host (sage_debug_varlen_items) and numpy."""
    import ctypes
    from sageattention_amd import _cabi
    lib = _cabi.load()
    g = torch.Generator().manual_seed(5)
    for nseq in (1, 2, 7, 64, 333, 1024):
        lq = torch.randint(0, 5000 if nseq < 300 else 700, (nseq,), generator=g)
        lk = torch.randint(0, 5000 if nseq < 300 else 700, (nseq,), generator=g)
        lq[0] = lq[-1]                                                 # a tie
        cu_q = torch.nn.functional.pad(lq.cumsum(0), (1, 0)).to(torch.int32).to(DEV)
        cu_k = torch.nn.functional.pad(lk.cumsum(0), (1, 0)).to(torch.int32).to(DEV)
        for causal, hq, hkv in ((False, 8, 2), (True, 12, 4)):
            plan = sq.varlen_plan(cu_q, cu_k, want_q_blocks=True, total_q=int(lq.sum()), total_k=int(lk.sum()), is_causal=causal,
                                  Hq=hq, Hkv=hkv, head_dim=128)
            assert torch.equal(plan.cu_qs.cpu(), torch.nn.functional.pad(((lq + 127) // 128).cumsum(0), (1, 0)).to(torch.int32))
            assert torch.equal(plan.cu_ks.cpu(), torch.nn.functional.pad(((lk + 63) // 64).cumsum(0), (1, 0)).to(torch.int32))
            o = plan.order.cpu().long()
            assert sorted(o.tolist()) == list(range(nseq)) and (lq[o][:-1] >= lq[o][1:]).all()
            nslab = (lk + 511) // 512
            sf = plan.slab_first.cpu()
            assert torch.equal(sf[:nseq + 1], torch.nn.functional.pad(nslab.cumsum(0), (1, 0)).to(torch.int32))
            assert sf[nseq + 1] == sf[nseq] == sf[nseq + 2]                       # no rows outside the sequences here: the gap segments are empty
            assert torch.equal(plan.slab_seq.cpu()[:int(nslab.sum())], torch.repeat_interleave(torch.arange(nseq), nslab).to(torch.int32))
            hdr = plan.hdr.cpu().numpy()
            nitems = int(((lq + 127) // 128).sum())
            assert hdr[0] == nitems <= plan.items_bound and hdr[4] == int(nslab.sum()) <= plan.slab_bound
            assert hdr[5] == int(lk.max()) and hdr[6] == int(lk.sum())
            lqa, lka = lq.numpy().astype(np.int32), lk.numpy().astype(np.int32)
            items = np.zeros((max(nitems, 1), 2), np.int32)
            hh = np.zeros(8, np.int32)
            grid = lib.sage_debug_varlen_items(lqa.ctypes.data_as(ctypes.c_void_p), lka.ctypes.data_as(ctypes.c_void_p), nseq, int(causal), hq, hkv,
                                               128, 0, items.ctypes.data_as(ctypes.c_void_p), max(nitems, 1), hh.ctypes.data_as(ctypes.c_void_p))
            assert grid >= 0 and (hh[:4] == hdr[:4]).all()
            assert (plan.items.cpu().numpy()[:nitems] == items[:nitems]).all()
    assert sq.varlen_plan(torch.zeros(1026, dtype=torch.int32, device=DEV), torch.zeros(1026, dtype=torch.int32, device=DEV)) is None


_VARLEN_SETS = [([1, 127, 128, 129, 700, 64, 1000], None), ([512, 513, 2048, 5, 1536], None), ([300], None), ([4096, 100, 4000], None),
                ([1, 127, 128, 129, 700, 64, 1000], [5, 64, 200, 77, 1000, 640, 3])]


@pytest.mark.parametrize("dt,D,hkv", [(1, 128, 2), (0, 64, 4), (0, 128, 1)])
@pytest.mark.parametrize("smooth_k", [True, False])
def test_varlen_one_launch_prepass_is_bit_identical_to_the_sequence(dt, D, hkv, smooth_k):
    """sage_prepass_kv_varlen (K mean over all packed tokens + per-sequence INT8 K + fp16 V image, one launch, K and V read once) against
    channel_mean_packed(plan) + per_block_int8_varlen + prep_v_fp16_varlen: every output bit, ragged lengths, slabs of several sequences."""
    T = torch.float16 if dt == 0 else torch.bfloat16
    for lens, klens in _VARLEN_SETS:
        klens = klens or lens
        total = sum(klens)
        g = torch.Generator().manual_seed(77 + total)
        k = (torch.randn(total, hkv, D, generator=g) * 1.5 + 2.0 * torch.randn(1, hkv, D, generator=g)).to(T).to(DEV)
        v = torch.randn(total, hkv, D, generator=g).to(T).to(DEV)
        cu_q = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
        cu_k = torch.tensor([0] + list(np.cumsum(klens)), dtype=torch.int32, device=DEV)
        plan = sq.varlen_plan(cu_q, cu_k, total_q=sum(lens), total_k=total, Hq=hkv, Hkv=hkv, head_dim=D)
        assert sq.prepass_varlen_fused_ok(k, plan, max(klens), smooth_k)
        sync = torch.zeros(int(sq._cabi.load().sage_prepass_sync_words(1, hkv)), dtype=torch.int32, device=DEV)
        km1, k81, ks1, img1 = sq.prepass_kv_varlen(k, v, cu_k, plan, max(klens), smooth_k=smooth_k, sync=sync)
        assert sq.prepass_failed_heads(sync, 1, hkv) == 0
        km0 = sq.channel_mean_packed(k, cu_k, plan) if smooth_k else None
        _, _, k80, ks0, _, _ = sq.per_block_int8_varlen(None, k, cu_q, cu_k, max(lens), max(klens), km=km0, cu_ks=plan.cu_ks)
        img0 = sq.prep_v_fp16_varlen(v, cu_k, plan.cu_ks, max(klens), ntiles=img1.shape[0])
        torch.cuda.synchronize()
        nblk = int(plan.cu_ks[-1].item())
        if smooth_k:
            assert torch.equal(km1.view(torch.int16), km0.view(torch.int16))
            want = k.float().mean(dim=0, keepdim=True)              # and it IS the mean (one rounding of an fp32 sum)
            assert (km1.float() - want).abs().max().item() <= 2.0 ** (-7 if dt == 1 else -10) * max(1.0, want.abs().max().item())
        assert torch.equal(k81, k80) and torch.equal(ks1[:nblk].view(torch.int32), ks0[:nblk].view(torch.int32))
        assert torch.equal(img1[:nblk].view(torch.int16), img0[:nblk].view(torch.int16))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("dt,D,hq,hkv", [(1, 128, 8, 2), (0, 64, 12, 4), (0, 128, 3, 1)])
def test_varlen_routes_are_bit_identical(causal, dt, D, hq, hkv):
    """sageattn_varlen's default route (plan launch + one-launch K / V pre-pass + attention over the device-built work list) against the
    kernel sequence, against the unit order sized by max_seqlen_q, and against both: same bits; and no host synchronisation."""
    T = torch.float16 if dt == 0 else torch.bfloat16
    for lens, klens in _VARLEN_SETS:
        if causal and klens is not None:
            continue
        klens = klens or lens
        g = torch.Generator().manual_seed(1234 + sum(lens))
        q = torch.randn(sum(lens), hq, D, generator=g).to(T).to(DEV)
        k = (torch.randn(sum(klens), hkv, D, generator=g) + torch.randn(1, hkv, D, generator=g)).to(T).to(DEV)
        v = torch.randn(sum(klens), hkv, D, generator=g).to(T).to(DEV)
        cu_q = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
        cu_k = torch.tensor([0] + list(np.cumsum(klens)), dtype=torch.int32, device=DEV)
        args = (q, k, v, cu_q, cu_k, max(lens), max(klens))
        o = sa.sageattn_varlen(*args, is_causal=causal)
        o_seq = sa.sageattn_varlen(*args, is_causal=causal, fused_prepass=False)
        o_unit = sa.sageattn_varlen(*args, is_causal=causal, work_list=False)
        o_both = sa.sageattn_varlen(*args, is_causal=causal, fused_prepass=False, work_list=False, fuse_q_quant=False)
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all()
        assert torch.equal(o, o_seq) and torch.equal(o, o_unit) and torch.equal(o, o_both)


def test_varlen_rows_outside_every_sequence_still_count_in_the_k_mean():
    """`km = k.mean(dim=0)` (core.py:432-434) averages over EVERY row of the packed k, also rows that belong to no sequence (a packed tensor
    padded behind cu_seqlens[-1], or cu_seqlens[0] > 0).  The per-sequence slabs of the plan are joined by two gap segments that only the
    statistics read: the mean is the mean over all rows on both routes, bit-equal between them, and the attention result equals the
    result of the same call on the tight tensors with that mean."""
    g = torch.Generator().manual_seed(31)
    head, lens, tail = 700, [300, 1100, 64, 513], 900
    total = head + sum(lens) + tail
    hq, hkv, D = 4, 2, 128
    q = torch.randn(total, hq, D, generator=g).to(torch.bfloat16).to(DEV)
    k = (torch.randn(total, hkv, D, generator=g) * 1.3 + torch.randn(1, hkv, D, generator=g)).to(torch.bfloat16).to(DEV)
    k[:head] += 3.0                                                  # the gap rows move the mean visibly
    k[-tail:] -= 2.0
    v = torch.randn(total, hkv, D, generator=g).to(torch.bfloat16).to(DEV)
    cu = torch.tensor([head + x for x in [0] + list(np.cumsum(lens))], dtype=torch.int32, device=DEV)
    plan = sq.varlen_plan(cu, cu, total_q=total, total_k=total, Hq=hq, Hkv=hkv, head_dim=D)
    hdr = plan.hdr.cpu().numpy()
    nslab_seq = sum((x + 511) // 512 for x in lens)
    assert hdr[4] == nslab_seq + (tail + 511) // 512 + (head + 511) // 512 and hdr[7] == hdr[4] - nslab_seq
    km1, k81, ks1, img1 = sq.prepass_kv_varlen(k, v, cu, plan, max(lens))
    km0 = sq.channel_mean_packed(k, cu, plan)
    want = k.float().mean(dim=0, keepdim=True)
    torch.cuda.synchronize()
    assert torch.equal(km1.view(torch.int16), km0.view(torch.int16))
    assert (km1.float() - want).abs().max().item() <= 2.0 ** -7 * max(1.0, want.abs().max().item())
    assert (km1.float() - k[head:head + sum(lens)].float().mean(dim=0, keepdim=True)).abs().max().item() > 0.05     # (not the mean of the sequences alone)
    _, _, k80, ks0, _, _ = sq.per_block_int8_varlen(None, k, cu, cu, max(lens), max(lens), km=km0, cu_ks=plan.cu_ks)
    nblk = int(plan.cu_ks[-1].item())
    lo, hi = head, head + sum(lens)
    assert torch.equal(k81[lo:hi], k80[lo:hi]) and torch.equal(ks1[:nblk].view(torch.int32), ks0[:nblk].view(torch.int32))
    for causal in (False, True):
        o = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
        o_seq = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal, fused_prepass=False)
        torch.cuda.synchronize()
        assert torch.isfinite(o[lo:hi].float()).all() and torch.equal(o[lo:hi], o_seq[lo:hi])


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_with_empty_sequences(causal):
    """Sequences of length zero (repeated entries of cu_seqlens; the reference's grid lets their blocks exit, attn_qk_int8_block_varlen.py:98-121):
    no work item, no slab, no scale block.  The call must equal, bit for bit, the call on the same packed tensors with the empty sequences
    removed from cu_seqlens -- on the planned route, on the kernel sequence and on the unit order."""
    g = torch.Generator().manual_seed(5)
    lens_full = [0, 200, 0, 0, 1025, 64, 0, 513, 0]
    lens = [x for x in lens_full if x > 0]
    hq, hkv, D = 8, 2, 128
    total = sum(lens)
    q = torch.randn(total, hq, D, generator=g).to(torch.bfloat16).to(DEV)
    k = (torch.randn(total, hkv, D, generator=g) + torch.randn(1, hkv, D, generator=g)).to(torch.bfloat16).to(DEV)
    v = torch.randn(total, hkv, D, generator=g).to(torch.bfloat16).to(DEV)
    cu_full = torch.tensor([0] + list(np.cumsum(lens_full)), dtype=torch.int32, device=DEV)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    want = sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens), is_causal=causal)
    for kw in ({}, {"fused_prepass": False}, {"work_list": False}, {"fused_prepass": False, "work_list": False, "fuse_q_quant": False}):
        o = sa.sageattn_varlen(q, k, v, cu_full, cu_full, max(lens), max(lens), is_causal=causal, **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all() and torch.equal(o, want), kw
    # keys but no queries / queries but no keys in one sequence (cu_seqlens_q != cu_seqlens_k): rows without keys are zero, as the
    # reference's accumulators are (acc = 0, l_i = 1: attn_qk_int8_block_varlen.py:57-58)
    if not causal:
        lq, lk = [128, 0, 300, 70], [200, 64, 0, 70]
        q2 = torch.randn(sum(lq), hq, D, generator=g).to(torch.bfloat16).to(DEV)
        k2 = torch.randn(sum(lk), hkv, D, generator=g).to(torch.bfloat16).to(DEV)
        v2 = torch.randn(sum(lk), hkv, D, generator=g).to(torch.bfloat16).to(DEV)
        cq = torch.tensor([0] + list(np.cumsum(lq)), dtype=torch.int32, device=DEV)
        ck = torch.tensor([0] + list(np.cumsum(lk)), dtype=torch.int32, device=DEV)
        o = sa.sageattn_varlen(q2, k2, v2, cq, ck, max(lq), max(lk))
        o_seq = sa.sageattn_varlen(q2, k2, v2, cq, ck, max(lq), max(lk), fused_prepass=False, work_list=False)
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all() and torch.equal(o, o_seq)
        assert (o[128:428] == 0).all() and (o[:128] != 0).any()


def test_persistent_launches_are_bit_identical_and_really_taken(monkeypatch):
    """Large non-causal calls hand the attention launch a zeroed counter block (SageLaunchAttr.launch_ws) and run as persistent launches:
    fewer workgroups than work items, the same output bits as the ordinary launch; causal calls and small calls stay ordinary launches; the
    attribute is an argument of its call -- a call without it is an ordinary launch whatever came before."""
    import ctypes
    from sageattention_amd import ops, _stream_cache as sc
    probe = ctypes.c_int32(-1)
    g = torch.Generator().manual_seed(77)

    def run(fn, on):
        monkeypatch.setattr(ops, "_PERSISTENT", on)
        probe.value = -1
        with ops.launch_hooks(grid_probe=probe):
            out = fn()
        torch.cuda.synchronize()
        if on:      # the stream's ticket block (if the call took one) is back at zero: the last workgroup to leave re-arms it, no memset per call
            blk = sc._CACHE.get(sc._key("attn_tickets", DEV))
            assert blk is None or int(blk.abs().max().item()) == 0, "a persistent launch must leave its counter block zero"
        return out, int(probe.value)

    # dense, FP8 PV (sageattn) at D = 64 (three workgroups per CU: 768 at once) and D = 128 (512), FP16 PV and the Triton-named API:
    # twelve rounds of workgroups = 2 * 36 * 128 = 9216 / 2 * 24 * 128 = 6144 work items
    for D, H, api in ((64, 36, sa.sageattn), (128, 24, sa.sageattn), (128, 24, sa.sageattn_qk_int8_pv_fp16_cuda),
                      (128, 24, sa.sageattn_qk_int8_pv_fp16_triton)):
        q, k, v = (torch.randn(2, H, 16384, D, generator=g).to(torch.bfloat16).to(DEV) for _ in range(3))
        o1, g1 = run(lambda: api(q, k, v, is_causal=False), True)
        o0, g0 = run(lambda: api(q, k, v, is_causal=False), False)
        assert g0 == 2 * H * 128 and 0 < g1 < g0 and g1 % 32 == 0, (api.__name__, D, g0, g1)
        assert torch.equal(o1, o0), (api.__name__, D)
        del o1, o0
        qc, kc, vc = q[:, :, :4096].contiguous(), k[:, :, :4096].contiguous(), v[:, :, :4096].contiguous()
        _, gc = run(lambda: api(qc, kc, vc, is_causal=True), True)            # causal: the hardware's dispatch
        assert gc >= 2 * H * 32
        _, gs = run(lambda: api(qc, kc, vc, is_causal=False), True)           # eight rounds at most: an ordinary launch
        assert gs == 2 * H * 32
        del q, k, v, qc, kc, vc
    # a launch without the attribute is an ordinary one (nothing of the previous call's attribute is left anywhere)
    q, k, v = (torch.randn(2, 24, 16384, 128, generator=g).to(torch.bfloat16).to(DEV) for _ in range(3))
    _, g1 = run(lambda: sa.sageattn(q, k, v), True)
    orig_ws = ops.attn_launch_ws
    monkeypatch.setattr(ops, "attn_launch_ws", lambda *a, **kw: None)
    _, g0 = run(lambda: sa.sageattn(q, k, v), True)
    assert g0 == 6144 and g1 < 6144
    monkeypatch.setattr(ops, "attn_launch_ws", orig_ws)
    del q, k, v
    # packed batches, non-causal: 16 heads x 52000 rows
    lens = [19000, 300, 17000, 129, 15571]
    q = torch.randn(sum(lens), 16, 128, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(sum(lens), 4, 128, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(sum(lens), 4, 128, generator=g).to(torch.bfloat16).to(DEV)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    o1, g1 = run(lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens)), True)
    o0, g0 = run(lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens)), False)
    assert 0 < g1 < g0 and torch.equal(o1, o0)
    # one block per stream serves every launch: the same tensor twice, zero in between, same bits; another stream gets another block
    blk = sc._CACHE[sc._key("attn_tickets", DEV)]
    o2, g2 = run(lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens)), True)
    assert sc._CACHE[sc._key("attn_tickets", DEV)] is blk and g2 == g1 and torch.equal(o2, o1)
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        o3, g3 = run(lambda: sa.sageattn_varlen(q, k, v, cu, cu, max(lens), max(lens)), True)
        assert sc._CACHE[sc._key("attn_tickets", DEV)] is not blk
    assert g3 == g1 and torch.equal(o3, o1)


def test_varlen_plan_clamps_counts_to_its_outputs_capacities():
    """sage_varlen_plan derives the item and slab counts from cu_seqlens ON THE DEVICE, while its output buffers are sized on the host from
    the packed row counts.  A cu_seqlens that is inconsistent with those rows (last prefix far beyond them) must not write past the buffers:
    the counts are clamped to the capacities handed over (hdr reports the clamped counts), guard words behind the buffers stay intact."""
    lib = _cabi.load()
    nseq, total = 3, 1000
    cu = torch.tensor([0, 300, 700, 200000], dtype=torch.int32, device=DEV)          # claims 200000 rows; the tensors have 1000
    items_cap = (total + 127) // 128 + nseq
    slab_cap = (total + 511) // 512 + nseq + 2
    GUARD = 0x5A5A5A5A
    items = torch.full((2 * items_cap + 64,), GUARD, dtype=torch.int32, device=DEV)
    slab_seq = torch.full((slab_cap + 64,), GUARD, dtype=torch.int32, device=DEV)
    cu_ks = torch.empty(nseq + 1, dtype=torch.int32, device=DEV)
    slab_first = torch.empty(nseq + 3, dtype=torch.int32, device=DEV)
    hdr = torch.empty(8, dtype=torch.int32, device=DEV)
    rc = lib.sage_varlen_plan(cu.data_ptr(), cu.data_ptr(), nseq, total, 128, 64, 1, 8, 4, 128, 0, None, cu_ks.data_ptr(), None,
                              items.data_ptr(), items_cap, slab_first.data_ptr(), slab_seq.data_ptr(), slab_cap, hdr.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "sage_varlen_plan")
    torch.cuda.synchronize()
    h = hdr.cpu().numpy()
    assert 0 < h[0] <= items_cap and 0 < h[4] <= slab_cap, h
    assert (items[2 * items_cap:] == GUARD).all() and (slab_seq[slab_cap:] == GUARD).all()
    it = items[:2 * int(h[0])].view(-1, 2).cpu().numpy()
    assert ((it[:, 0] >= 0) & (it[:, 0] < nseq)).all() and (slab_seq[:int(h[4])].cpu().numpy() < nseq + 2).all()
    # a consistent batch is untouched by the clamp: the plan of the same lengths with true row counts
    ok = sq.varlen_plan(torch.tensor([0, 300, 700, 1000], dtype=torch.int32, device=DEV), torch.tensor([0, 300, 700, 1000], dtype=torch.int32, device=DEV),
                        total_q=total, total_k=total, is_causal=True, Hq=8, Hkv=4)
    torch.cuda.synchronize()
    assert int(ok.hdr[0].item()) == 3 + 4 + 3 and int(ok.hdr[4].item()) == 1 + 1 + 1


def test_varlen_with_more_sequences_than_the_plan_takes(oracle_mod):
    """More than sage_varlen_plan_max_seqs() sequences: no plan, so torch prefix sums, an on-device argsort for the unit order, the kernel
    sequence, the Q quantiser fused in the attention prologue -- exercised end to end (round 3 only checked that the planner returns None).
    smooth_k=False makes the sequences independent of each other: the call must equal, bit for bit, two planned calls over its halves;
    smooth_k=True is checked against the oracle."""
    g = torch.Generator().manual_seed(99)
    nseq = 1100
    lens = torch.randint(1, 150, (nseq,), generator=g)
    lens[7] = 700
    hq, hkv, D = 4, 2, 64
    total = int(lens.sum())
    q = torch.randn(total, hq, D, generator=g).half()
    k = (torch.randn(total, hkv, D, generator=g) + torch.randn(1, hkv, D, generator=g)).half()
    v = torch.randn(total, hkv, D, generator=g).half()
    cu = torch.nn.functional.pad(lens.cumsum(0), (1, 0)).to(torch.int32)
    qd, kd, vd, cud = q.to(DEV), k.to(DEV), v.to(DEV), cu.to(DEV)
    assert sq.varlen_plan(cud, cud, total_q=total, total_k=total) is None
    for causal in (False, True):
        o = sa.sageattn_varlen(qd, kd, vd, cud, cud, int(lens.max()), int(lens.max()), is_causal=causal, smooth_k=False)
        half = 550
        t0 = int(cu[half])
        o_a = sa.sageattn_varlen(qd[:t0], kd[:t0], vd[:t0], cud[:half + 1].contiguous(), cud[:half + 1].contiguous(), int(lens[:half].max()),
                                 int(lens[:half].max()), is_causal=causal, smooth_k=False)
        cub = (cud[half:] - cud[half]).contiguous()
        o_b = sa.sageattn_varlen(qd[t0:], kd[t0:], vd[t0:], cub, cub, int(lens[half:].max()), int(lens[half:].max()), is_causal=causal, smooth_k=False)
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all() and torch.equal(o[:t0], o_a) and torch.equal(o[t0:], o_b)
    o = sa.sageattn_varlen(qd, kd, vd, cud, cud, int(lens.max()), int(lens.max()), is_causal=True)
    torch.cuda.synchronize()
    km = util.bits(sq.channel_mean_packed(kd))                      # no plan: the packed-token partition
    ref = oracle_mod.sageattn_varlen(util.bits(q), util.bits(k), util.bits(v), 0, cu.numpy(), cu.numpy(), is_causal=True, km=km)
    _assert_vs_oracle("varlen_1100_sequences_causal", o.float().cpu().numpy(), ref, 0)


def test_graphed_sageattn_replays_bit_identically_and_cuts_host_time():
    import time
    from sageattention_amd.graph import GraphedSageAttn
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 8, 8, 256, 256, 128, 1, seed=9))
    ga = GraphedSageAttn(q, k, v, is_causal=True, return_lse=True)
    q2, k2, v2 = (t.to(DEV) for t in rand_qkv(1, 8, 8, 256, 256, 128, 1, seed=10))
    o, lse = ga(q2, k2, v2)
    want_o, want_lse = sa.sageattn(q2, k2, v2, is_causal=True, return_lse=True)
    torch.cuda.synchronize()
    assert torch.equal(o, want_o) and torch.equal(lse, want_lse)
    def host_us(fn, n=40, batches=6):
        # the best of several short batches: a batch that runs into a full queue (the host then waits for the GPU) or into a busy host
        # measures something else -- one 200-call batch per route failed on a slow box (graphed 282 us against 25 us on every other box)
        best = float("inf")
        for _ in range(batches):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                fn()
            best = min(best, (time.perf_counter() - t0) / n * 1e6)
        torch.cuda.synchronize()
        return best
    eager, graphed = host_us(lambda: sa.sageattn(q2, k2, v2, is_causal=True, return_lse=True)), host_us(ga.replay)
    REPORT["graph/host_us_per_call"] = dict(eager=eager, graphed=graphed)
    assert graphed < eager


def test_custom_ops_pass_opcheck():
    """torch.library.opcheck on the two registered ops: schema (mutates `output` only), fake-tensor kernel agreement."""
    from sageattention_amd import _cabi, ops
    q, k, v = (t.to(DEV) for t in rand_qkv(1, 2, 2, 256, 192, 128, 0, seed=23))
    km = sq.channel_mean(k)
    q8, qs, k8, ks = sq.per_thread_int8(q, k, km)
    img8, vs, _ = sq.per_channel_fp8(v)
    img16 = sq.prep_v_fp16(v)
    o = torch.empty_like(q)
    for return_lse in (0, 1):
        torch.library.opcheck(ops.qk_int8_sv_f8_attn, (q8, k8, img8, o, qs, ks, vs, None, 1, 1, _cabi.GRAN_PER_THREAD, 32, 0.1275, _cabi.PV_ACCUM_TWO_LEVEL, return_lse),
                              test_utils=("test_schema", "test_faketensor"))
        torch.library.opcheck(ops.qk_int8_sv_f16_attn, (q8, k8, img16, o, qs, ks, None, 1, 0, _cabi.GRAN_PER_THREAD, 32, 0.1275, _cabi.PV_ACCUM_SINGLE, return_lse),
                              test_utils=("test_schema", "test_faketensor"))
