"""The C oracle's FP8-PV path against a second, independently written restatement (tests/ref_fp8_torch.py).

The reference's FP8 kernels are CUDA-only, so no reference output exists to pin the oracle to; instead the algorithm was
restated twice from the reference text -- once in C (oracle/sage_oracle.c, scalar loops), once in PyTorch-CPU
(tile-vectorised, torch.float8_e4m3fn casts) -- and the two must agree: every integer / byte / scale tensor
bit-for-bit, the fp16/bf16 output to the last ulp (the two differ only in libm's exp2f/log2f against correctly
rounded fp64 evaluations), the LSE to 2e-6.  CPU only.
"""
import numpy as np
import pytest
import torch

import oracle
from tests import ref_fp8_torch as ref


def _bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def _mk(shape, dtype, seed, bias=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    if bias:
        x = x + bias * torch.randn(shape[:-2] + (1, shape[-1]), generator=g)
    return x.to(dtype)


CASES = [
    # B, Hq, Hkv, Lq, Lk, D, dtype, causal, gran, accum, smooth_v
    (1, 2, 1, 200, 200, 64, torch.float16, True, "per_thread", "fp32+fp32", False),
    (1, 2, 2, 150, 333, 128, torch.bfloat16, False, "per_warp", "fp32+fp32", False),
    (2, 4, 2, 129, 129, 128, torch.float16, True, "per_warp", "fp32", True),
    (1, 1, 1, 77, 300, 64, torch.bfloat16, False, "per_thread", "fp32", True),
    (1, 2, 1, 256, 256, 128, torch.float16, True, "per_thread", "fp32", False),
]


@pytest.mark.parametrize("B,Hq,Hkv,Lq,Lk,D,dtype,causal,gran,accum,smooth_v", CASES)
def test_oracle_equals_torch_restatement(B, Hq, Hkv, Lq, Lk, D, dtype, causal, gran, accum, smooth_v):
    q = _mk((B, Hq, Lq, D), dtype, 1)
    k = _mk((B, Hkv, Lk, D), dtype, 2, bias=3.0)        # per-channel K bias: smoothing matters
    v = _mk((B, Hkv, Lk, D), dtype, 3, bias=1.0 if smooth_v else 0.0)
    o_t, lse_t, aux_t = ref.sageattn_fp8_cuda(q, k, v, is_causal=causal, qk_quant_gran=gran, smooth_v=smooth_v,
                                              pv_accum_dtype=accum)
    code = oracle.F16 if dtype == torch.float16 else oracle.BF16
    # the oracle's dense API computes two-level accumulation; the single-level variant goes through attn() directly
    o_c, lse_c, aux_c = oracle.sageattn_dense(_bits(q), _bits(k), _bits(v), code, is_causal=causal, qk_quant_gran=gran,
                                              pv="f8", return_lse=True, smooth_k=True, smooth_v=smooth_v)
    # ---- quantised operands: bit-for-bit
    assert np.array_equal(aux_c["q8"], aux_t["q8"].numpy()), "q int8"
    assert np.array_equal(aux_c["k8"], aux_t["k8"].numpy()), "k int8"
    assert np.array_equal(aux_c["qs"].view(np.uint32), aux_t["qs"].numpy().view(np.uint32)), "q scale"
    assert np.array_equal(aux_c["ks"].view(np.uint32), aux_t["ks"].numpy().view(np.uint32)), "k scale"
    assert np.array_equal(aux_c["km"], _bits(aux_t["km"])), "k mean"
    assert np.array_equal(aux_c["v8"], aux_t["v8"].numpy()), "v e4m3 bytes"
    assert np.array_equal(aux_c["vs"].view(np.uint32), aux_t["vs"].numpy().view(np.uint32)), "v scale"
    if smooth_v:
        assert np.array_equal(aux_c["vm"].view(np.uint32), aux_t["vm"].numpy().view(np.uint32)), "v mean"
    if accum == "fp32":      # single-level accumulation (sm89 "accum_f32" kernels, no instruction buffer)
        o_c, lse_c = oracle.attn(aux_c["q8"], aux_c["k8"], aux_c["v8"], aux_c["qs"], aux_c["gq"], aux_c["ks"], aux_c["gk"],
                                 causal=causal, c=aux_c["c"], pv_mode=oracle.PV_F8_SINGLE, out_dtype=code,
                                 v_scale=aux_c["vs"], v_mean=aux_c["vm"] if smooth_v else None, return_lse=True)
    else:
        lse_c = None
    # ---- output: identical up to the last ulp of the output dtype on a vanishing fraction of elements
    got, want = o_c.astype(np.int32), _bits(o_t).astype(np.int32)
    diff = np.abs(got - want)
    assert diff.max() <= 1, f"max ulp distance {diff.max()}"
    assert (diff != 0).mean() < 2e-3, f"{(diff != 0).mean():.2e} of the outputs differ by one ulp"
    if lse_c is not None:
        assert np.abs(lse_c - lse_t.numpy()).max() < 2e-6 * max(1.0, float(np.abs(lse_c).max()))


def test_smooth_v_padding_zeros_enter_the_statistics():
    """fused.cu:335-357: MeanScaleKernel scans ceil16(L) tokens of the zero-padded transpose, so with smooth_v and
    L % 16 != 0 the amax includes |0 - mean|.  A channel whose values all sit far from zero shows the difference."""
    L, D = 21, 64
    v = (5.0 + 0.01 * torch.randn(1, 1, L, D, generator=torch.Generator().manual_seed(0))).to(torch.float16)
    vb = _bits(v)
    vm = oracle.v_mean_padded16(vb, oracle.F16)
    v8, vs = oracle.quant_v_fp8(vb, oracle.F16, mean=vm)
    t8, ts, tm = ref.quant_v_fp8(v[0, 0], True)
    assert np.array_equal(vm[0, 0].view(np.uint32), tm.numpy().view(np.uint32))
    assert np.array_equal(vs[0, 0].view(np.uint32), ts.numpy().view(np.uint32))
    assert np.array_equal(v8[0, 0], t8.view(torch.uint8).numpy())
    # mean = 5 * 21/32; the padding zeros sit |mean| ~ 3.3 away, the real tokens only ~1.7: amax must be the former
    assert np.all(vs[0, 0] * 448.0 > 3.0)


@pytest.mark.parametrize("Lq,Lk,D,dtype,causal,gran", [(200, 333, 64, torch.float16, False, "per_warp"), (150, 150, 128, torch.bfloat16, True, "per_thread"),
                                                     (129, 700, 128, torch.float16, False, "per_thread")])
def test_fp16_pv_cuda_form_denominator_sums_the_rounded_p(Lq, Lk, D, dtype, causal, gran):
    """oracle pv_mode 1 (the FP16-PV CUDA kernels) against the torch restatement of qk_int_sv_f16_cuda_sm80.cu: the softmax
    denominator is the FP32 sum of the fp16-ROUNDED probabilities (tensor-core row sum, sm80.cu:313-320).  The Lk = 333 / per-warp
    case is the one where an implementation that drops fp16-subnormal P from the sum is 0.6-1.7 % off (profiles/r3_run_d_*): the two
    restatements must agree to the last output ulp, and both must differ measurably from the sum of the un-rounded P."""
    q = _mk((1, 1, Lq, D), dtype, 11)
    k = _mk((1, 1, Lk, D), dtype, 12, bias=3.0)
    v = _mk((1, 1, Lk, D), dtype, 13)
    code = oracle.F16 if dtype == torch.float16 else oracle.BF16
    o_c, lse_c, aux = oracle.sageattn_dense(_bits(q), _bits(k), _bits(v), code, is_causal=causal, qk_quant_gran=gran, pv="f16",
                                            return_lse=True, smooth_k=True)
    gq, gk = torch.from_numpy(aux["gq"].astype("int64")), torch.from_numpy(aux["gk"].astype("int64"))
    vh = v[0, 0].to(torch.float16)
    o_t, lse_t = ref.attn_f16(torch.from_numpy(aux["q8"][0, 0]), torch.from_numpy(aux["k8"][0, 0]), vh,
                              torch.from_numpy(aux["qs"][0, 0]), gq, torch.from_numpy(aux["ks"][0, 0]), gk,
                              causal=causal, sm_scale=D ** -0.5, out_dtype=dtype)
    got, want = o_c[0, 0].astype(np.int32), _bits(o_t).astype(np.int32)
    diff = np.abs(got - want)
    assert diff.max() <= 1 and (diff != 0).mean() < 5e-3, (int(diff.max()), float((diff != 0).mean()))
    # the raw log-sum-exp of the kernel (log2 units, before the host's q.km correction): recompute it from the oracle's pieces
    o_raw, lse_raw = oracle.attn(aux["q8"], aux["k8"], _bits(vh)[None, None], aux["qs"], aux["gq"], aux["ks"], aux["gk"], causal=causal,
                                 c=aux["c"], pv_mode=oracle.PV_F16_F32ACC, out_dtype=code, return_lse=True)
    assert np.abs(lse_raw[0, 0] - lse_t.numpy()).max() < 2e-6 * max(1.0, float(np.abs(lse_t.numpy()).max()))
    # and the Triton form (un-rounded sum) is a different number: the two modes are told apart by their denominators
    _, lse_tr = oracle.attn(aux["q8"], aux["k8"], _bits(vh)[None, None], aux["qs"], aux["gq"], aux["ks"], aux["gk"], causal=causal,
                            c=aux["c"], pv_mode=oracle.PV_F16_TRITON, out_dtype=code, return_lse=True)
    assert np.abs(lse_tr[0, 0] - lse_raw[0, 0]).max() > 1e-6
