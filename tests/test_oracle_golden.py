"""CPU: the oracle against the golden vectors produced by the REFERENCE (Triton kernels under the
CPU interpreter, tests/golden/gen_golden.py) and against fp32 SDPA.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import util

DENSE = ["c1_b1h4n512d64_f16", "gqa_causal_n300d128_bf16", "cross_lq200_lk333_d64_f16", "causal_n384d128_f16", "pad_d96_n160_f16",
         "long_nc_lq256_lk1100_d128_f16", "long_c_n1000_d64_bf16"]       # (round 6: key ranges of several trips of the kernels' six-body loop)
VARLEN = ["varlen_nc_d64_f16", "varlen_c_d64_f16", "varlen_c_d128_bf16"]


def test_conversions_match_torch(oracle_mod):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(50000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 6e4)] +
                       [np.array([0, -0.0, 65504, 65519.99, 65520, 2 ** -24, 2 ** -25, 448, 447.9, 464, -448, 2 ** -9, 2 ** -10, 0.0146, 0.0156], dtype=np.float32)])
    t = torch.from_numpy(x)
    assert (oracle_mod.convert(x, "f16") == t.to(torch.float16).view(torch.int16).numpy().view(np.uint16)).all()
    assert (oracle_mod.convert(x, "bf16") == t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)).all()
    xc = np.clip(x, -448, 448)
    assert (oracle_mod.convert(xc, "e4m3") == torch.from_numpy(xc).to(torch.float8_e4m3fn).view(torch.uint8).numpy()).all()
    # saturate-to-finite above 448 (cvt.rn.satfinite), unlike torch which produces NaN
    assert oracle_mod.convert(np.array([1000.0, -1e9], dtype=np.float32), "e4m3").tolist() == [0x7E, 0xFE]


@pytest.mark.parametrize("name", DENSE)
def test_dense_matches_reference(oracle_mod, name):
    z, (B, Hq, Hkv, Lq, Lk, D, dt, causal) = util.golden(name)
    o, lse, aux = oracle_mod.sageattn_dense(z["q"], z["k"], z["v"], dt, is_causal=bool(causal), pv="f16_triton", return_lse=True)
    # integer work is bit-exact
    assert (aux["q8"] == z["q_int8"]).all() and (aux["k8"] == z["k_int8"]).all()
    assert (aux["qs"] == z["q_scale"]).all() and (aux["ks"] == z["k_scale"]).all()
    assert (aux["km"] == z["km"][:, :, 0, :]).all()
    of, rf = util.f32(o, dt), util.f32(z["o"], dt)
    # fp16: only summation order differs -> <= 1 ulp.  bf16: the CPU interpreter TRUNCATES fp32->bf16
    # (triton/runtime/interpreter.py _convert_float, rounding_mode=None) where real Triton and this
    # oracle round to nearest even -> up to 1 bf16 ulp (2^-7 relative).
    tol = (2 ** -10 if dt == 0 else 2 ** -7) * max(1.0, float(np.abs(rf).max()))
    assert np.abs(of - rf).max() <= tol
    assert np.abs(lse - z["lse"]).max() < 1e-5
    if dt == 0:
        assert (o == z["o"]).mean() > 0.995


@pytest.mark.parametrize("name", VARLEN)
def test_varlen_matches_reference(oracle_mod, name):
    z, (nseq, Hq, Hkv, total, _, D, dt, causal) = util.golden(name)
    o = oracle_mod.sageattn_varlen(z["q"], z["k"], z["v"], dt, z["cu"], z["cu"], is_causal=bool(causal))
    of, rf = util.f32(o, dt), util.f32(z["o"], dt)
    tol = (2 ** -10 if dt == 0 else 2 ** -7) * max(1.0, float(np.abs(rf).max()))
    assert np.abs(of - rf).max() <= tol


@pytest.mark.parametrize("name", ["varlenx_nc_d128_bf16", "varlenx_c_d64_f16"])
def test_varlen_cu_q_differs_from_cu_k_matches_reference(oracle_mod, name):
    """cu_seqlens_q != cu_seqlens_k (per-sequence Lq != Lk; causal top-left aligned) against the reference kernels."""
    z, (nseq, Hq, Hkv, tq, tk, D, dt, causal) = util.golden(name)
    assert not np.array_equal(z["cu_q"], z["cu_k"])
    o = oracle_mod.sageattn_varlen(z["q"], z["k"], z["v"], dt, z["cu_q"], z["cu_k"], is_causal=bool(causal))
    of, rf = util.f32(o, dt), util.f32(z["o"], dt)
    tol = (2 ** -10 if dt == 0 else 2 ** -7) * max(1.0, float(np.abs(rf).max()))
    assert np.abs(of - rf).max() <= tol


@pytest.mark.parametrize("name,kind", [("mask_bool_lq300_lk333_d64_f16", "bool"), ("mask_add_lq200_lk256_d128_bf16", "add"),
                                       ("mask_bool_skipall_lq140_lk130_d64_f16", "bool")])
def test_attn_mask_matches_reference(oracle_mod, name, kind):
    """Triton API attn_mask semantics (bool 0/-1e6 + all-False tile skip, additive float), incl. fully masked rows -- and (skipall) a query block
    whose EVERY tile is skipped: the reference's l_i starts at 1.0 (attn_qk_int8_per_block.py:112), so its rows are 0 with an lse of -inf."""
    z, (B, Hq, Hkv, Lq, Lk, D, dt, _) = util.golden(name)
    kw = dict(mask_bool=z["mask"]) if kind == "bool" else dict(mask_add=util.f32(z["mask"], dt))
    o, lse, _ = oracle_mod.sageattn_dense(z["q"], z["k"], z["v"], dt, pv="f16_triton", return_lse=True, **kw)
    of, rf = util.f32(o, dt), util.f32(z["o"], dt)
    tol = (2 ** -10 if dt == 0 else 2 ** -7) * max(1.0, float(np.abs(rf).max()))
    assert np.abs(of - rf).max() <= tol
    fin = np.isfinite(z["lse"])
    assert np.array_equal(np.isneginf(lse), np.isneginf(z["lse"])) and (name.find("skipall") < 0 or (~fin).sum() == B * Hq * 128)
    assert np.abs(lse[fin] - z["lse"][fin]).max() < 1e-4


def test_per_thread_quant_matches_reference(oracle_mod):
    z, (B, Hq, Hkv, Lq, Lk, D, dt, _) = util.golden("per_thread_quant_d128_f16")
    gq, nq = oracle_mod.group_index(Lq, "per_thread", "q", 128, 32)
    gk, nk = oracle_mod.group_index(Lk, "per_thread", "k", 64, 64)
    q8, qs = oracle_mod.quant_int8(z["q"], dt, gq, nq, style=oracle_mod.STYLE_TRITON_THREAD)
    k8, ks = oracle_mod.quant_int8(z["k"], dt, gk, nk, style=oracle_mod.STYLE_TRITON_THREAD, mean=np.ascontiguousarray(z["km"][:, :, 0, :]))
    assert (q8 == z["q_int8"]).all() and (k8 == z["k_int8"]).all()
    assert (qs == z["q_scale"]).all() and (ks == z["k_scale"]).all()


@pytest.mark.parametrize("name", ["per_thread_sm90_d128_f16", "per_thread_sm90_d64_bf16", "per_thread_warpq16_d128_f16"])
def test_per_thread_quant_groups_match_reference(oracle_mod, name):
    """The per-thread quantiser in the groups of the other callers -- the sm90 API (BLKQ 64, WARPQ 16, BLKK 128, WARPK 128: core.py:967) and the
    fp16+fp32 API at D = 128 (WARPQ 16: core.py:604) -- against the reference's own output (quant_per_thread.py:154-203 under the interpreter):
    every INT8 byte and every scale."""
    z, (B, Hq, Hkv, Lq, Lk, D, dt, _) = util.golden(name)
    BLKQ, WARPQ, BLKK, WARPK = (int(x) for x in z["groups"])
    gq, nq = oracle_mod.group_index(Lq, "per_thread", "q", BLKQ, WARPQ)
    gk, nk = oracle_mod.group_index(Lk, "per_thread", "k", BLKK, WARPK)
    q8, qs = oracle_mod.quant_int8(z["q"], dt, gq, nq, style=oracle_mod.STYLE_TRITON_THREAD)
    k8, ks = oracle_mod.quant_int8(z["k"], dt, gk, nk, style=oracle_mod.STYLE_TRITON_THREAD, mean=np.ascontiguousarray(z["km"][:, :, 0, :]))
    assert q8.shape == z["q_int8"].shape and qs.shape == z["q_scale"].shape and ks.shape == z["k_scale"].shape
    assert (q8 == z["q_int8"]).all() and (k8 == z["k_int8"]).all()
    assert (qs == z["q_scale"]).all() and (ks == z["k_scale"]).all()


@pytest.mark.parametrize("pv,gran", [("f16_triton", "per_block"), ("f16", "per_warp"), ("f8", "per_warp"), ("f8", "per_thread"), ("f8", "per_block")])
@pytest.mark.parametrize("causal", [False, True])
def test_accuracy_vs_fp32_sdpa(oracle_mod, pv, gran, causal):
    """Stated bounds vs fp32 SDPA on randn inputs with a large per-channel K bias (where smoothing
    matters): FP16 PV cos >= 0.9995; FP8 PV cos >= 0.9990 -- e4m3 P and V carry
    2^-4 relative rounding error each and randn V averages towards zero, so the cosine is lower at
    equal RMSE.  Relative RMSE (RMSE / RMS(truth)) <= 2% (FP16 PV) / 5% (FP8 PV); measured 1.1-1.3% / 3.3-3.8%."""
    torch.manual_seed(11)
    B, Hq, Hkv, L, D = 1, 4, 2, 320, 128
    q = torch.randn(B, Hq, L, D).half()
    k = (torch.randn(B, Hkv, L, D) + 4.0 * torch.randn(1, Hkv, 1, D)).half()
    v = torch.randn(B, Hkv, L, D).half()
    o, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 0, is_causal=causal, pv=pv, qk_quant_gran=gran)
    truth = util.sdpa_f32(q, k, v, causal).numpy()
    of = util.f32(o, 0)
    assert util.cos_sim(of, truth) >= (0.9990 if pv == "f8" else 0.9995)
    assert util.rmse(of, truth) / float(np.sqrt((truth ** 2).mean())) <= (0.05 if pv == "f8" else 0.02)


def test_two_level_equals_single_level_to_rounding(oracle_mod):
    torch.manual_seed(3)
    q = torch.randn(1, 2, 200, 64).half(); k = torch.randn(1, 2, 200, 64).half(); v = torch.randn(1, 2, 200, 64).half()
    _, _, aux = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), 0, pv="f8", qk_quant_gran="per_warp")
    outs = []
    for mode in (oracle_mod.PV_F8_TWO_LEVEL, oracle_mod.PV_F8_SINGLE):
        o, _ = oracle_mod.attn(aux["q8"], aux["k8"], aux["v8"], aux["qs"], aux["gq"], aux["ks"], aux["gk"], causal=False,
                               c=aux["c"], pv_mode=mode, out_dtype=0, v_scale=aux["vs"])
        outs.append(util.f32(o, 0))
    assert np.abs(outs[0] - outs[1]).max() <= 2 ** -10


@pytest.mark.parametrize("case", [(1, 4, 2, 300, 300, 128, 0, True, 1.0, "per_thread"), (1, 2, 2, 512, 512, 64, 0, False, 0.0, "per_thread"),
                                  (1, 2, 1, 1000, 1000, 128, 1, True, 5.0, "per_thread"), (2, 2, 2, 200, 333, 64, 1, False, 2.0, "per_warp"),
                                  (1, 2, 2, 130, 1100, 128, 0, False, 1.0, "per_block")],
                         ids=["n300_d128_causal", "n512_d64", "n1000_d128_biased_k_bf16", "cross_d64_per_warp", "long_kv_per_block"])
def test_folded_scores_vs_exact(oracle_mod, case):
    """The oracle's two FP8 score forms on the CPU.  `exact` is the reference's formula exp2(fma(s, c, -m)) (attn_utils.cuh:445-449) and
    stays the pinned mode; `folded` is its reassociation exp2(fma(bits, c', -(m + bias c'))) as the gfx950 kernels' opt-in FP8 variant
    evaluates it.  What the variant is held to besides its own oracle mode: rel-RMS <= 1e-2 between the two forms on ordinary
    inputs, and accuracy against fp32 SDPA within 1e-4 (cos) / 1e-3 (rel-RMSE) of the exact form's.  The forms are NOT equal: single outputs of
    rows of a few hundred keys differ by up to ~1.5e-2 * max|o| (an e4m3 rounding of a large P flips), which is why each form has its own
    2e-3 gate instead of one gate for both."""
    B, Hq, Hkv, Lq, Lk, D, dt, causal, kbias, gran = case
    g = torch.Generator().manual_seed(40 + Lq)
    T = torch.float16 if dt == 0 else torch.bfloat16
    q = torch.randn(B, Hq, Lq, D, generator=g).to(T)
    k = (torch.randn(B, Hkv, Lk, D, generator=g) + kbias * torch.randn(1, Hkv, 1, D, generator=g)).to(T)
    v = torch.randn(B, Hkv, Lk, D, generator=g).to(T)
    outs = {}
    for form in ("exact", "folded"):
        for single in (False, True):
            o, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8", qk_quant_gran=gran,
                                                fp8_scores=form, single_level=single)
            outs[form, single] = util.f32(o, dt)
    truth = util.sdpa_f32(q, k, v, causal).numpy()
    tn = float(np.sqrt((truth ** 2).mean()))
    for single in (False, True):
        a, b = outs["folded", single], outs["exact", single]
        assert not np.array_equal(a, b) or Lk <= 64
        rel_rms = float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))
        assert rel_rms <= 1e-2, rel_rms
        assert np.abs(a - b).max() <= 4e-2 * np.abs(b).max()
        assert abs(util.cos_sim(a, truth) - util.cos_sim(b, truth)) <= 1e-4
        assert abs(util.rmse(a, truth) - util.rmse(b, truth)) / tn <= 1e-3
    # the folded form exists for the FP8 modes only
    _, _, aux = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16", qk_quant_gran=gran)
    with pytest.raises(AssertionError):
        oracle_mod.attn(aux["q8"], aux["k8"], util.bits(v.half()), aux["qs"], aux["gq"], aux["ks"], aux["gk"], causal=causal, c=aux["c"],
                        pv_mode=oracle_mod.PV_F16_F32ACC, out_dtype=dt, score_mode=oracle_mod.SCORES_FOLDED)


@pytest.mark.parametrize("case", [(300, 128, True, 1), (1000, 64, False, 1), (777, 128, False, 0)], ids=["n300_d128_causal_bf16", "n1000_d64_bf16", "n777_d128_f16"])
def test_sm90_tile_schedule_64_vs_128_keys(oracle_mod, case):
    """The reference's sm90 kernel iterates over 128 keys at a time -- one maximum update, one `RO += RO_temp` per 128 keys
    (qk_int_sv_f8_cuda_sm90.cu:285-356) -- where its sm89 kernel, its Triton kernels and every gfx950 kernel take 64.  The two schedules are the same
    arithmetic up to which maximum the first 64 keys of a tile are rounded to e4m3 against (and the FP32 summation order).  The oracle restates both
    (tile_keys); this pins how far apart they are, so that the sm90-named entry point's divergence from the reference's own sm90 schedule has a
    number: rel-RMS <= 1.5e-2, single outputs <= 3e-2 * max|o| (measured 0.8-1.0e-2 / 0.4-1.2e-2 -- the size of the e4m3 noise itself), and the same
    accuracy against fp32 SDPA to 2e-4 (cos) / 2e-3 (rel-RMSE)."""
    L, D, causal, dt = case
    g = torch.Generator().manual_seed(70 + L)
    T = torch.float16 if dt == 0 else torch.bfloat16
    q = torch.randn(1, 4, L, D, generator=g).to(T)
    k = (torch.randn(1, 2, L, D, generator=g) + torch.randn(1, 2, 1, D, generator=g)).to(T)
    v = torch.randn(1, 2, L, D, generator=g).to(T)
    o = {}
    for tk in (64, 128):
        ob, _, _ = oracle_mod.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8", qk_quant_gran="per_thread",
                                             warpq=16, blkk=128, tile_keys=tk)            # the sm90 API's scale groups (core.py:964-970)
        o[tk] = util.f32(ob, dt)
    d = o[64] - o[128]
    rel_rms = float(np.sqrt((d ** 2).mean() / (o[128] ** 2).mean()))
    assert rel_rms <= 1.5e-2 and np.abs(d).max() <= 3e-2 * np.abs(o[128]).max(), (rel_rms, np.abs(d).max())
    truth = util.sdpa_f32(q, k, v, causal).numpy()
    tn = float(np.sqrt((truth ** 2).mean()))
    acc = {tk: (util.cos_sim(o[tk], truth), util.rmse(o[tk], truth) / tn) for tk in o}
    assert abs(acc[64][0] - acc[128][0]) <= 2e-4 and abs(acc[64][1] - acc[128][1]) <= 2e-3, acc
