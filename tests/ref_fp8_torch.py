"""Second, independently written CPU restatement of the reference's FP8-PV CUDA path -- TEST INFRASTRUCTURE ONLY.

The reference's INT8-QK / FP8-PV kernels exist only as CUDA + PTX (no nvcc in this image), so the C oracle
(``oracle/sage_oracle.c``) cannot be pinned against their outputs.  This module restates the same algorithm a second
time, in a different language and with a different structure (tile-vectorised PyTorch-CPU, ``torch.float8_e4m3fn``
casts), straight from the reference sources; ``tests/test_second_restatement.py`` asserts that the two restatements
agree (integer / byte tensors bit-for-bit, outputs to the last ulp of the output dtype).  Two restatements written
from the text agreeing is the strongest pin available without a CUDA device.

Reference text followed (relative to /root/reference):
  INT8 quantisation, CUDA rounding   csrc/fused/fused.cu:110-186, numeric_conversion.cuh:144-149
  per-warp grouping                  sageattention/quant.py:169-178
  per-thread quantisation            sageattention/triton/quant_per_thread.py:21-98
  FP8 V quantisation (+ smooth_v)    sageattention/quant.py:269-293, csrc/fused/fused.cu:262-313 (zero padding), :316-427
  score scale                        csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:263-266,334-335
  online softmax with exp offset     csrc/qattn/attn_utils.cuh:30-32,354-458
  masks                              csrc/qattn/attn_utils.cuh:296-351
  P -> e4m3                          csrc/qattn/attn_utils.cuh:478-493
  row sum of the fp32 P              csrc/qattn/attn_utils.cuh:529-560 (ComputeUnit::kCudaCore, sm89.cuh:316)
  two-level accumulation             csrc/qattn/attn_utils.cuh:813-894
  epilogue                           csrc/qattn/qk_int_sv_f8_cuda_sm89.cuh:572-656, LSE :691-703
Nothing here imports ``oracle`` or ``sageattention_amd``.

Divisions are IEEE (correctly rounded) in both restatements.  The reference itself is built with ``--use_fast_math``
(setup.py:57), so its own quotients (127/amax, scale_max/amax, O/l) are only within ~2 ulp of these.
"""
from __future__ import annotations

import torch

S_FP8_OFFSET = 8.807            # attn_utils.cuh:30
LOG2E = 1.44269504088896340736  # csrc/math.cuh:32, applied in fp32 inside the kernel (qk_int_sv_f8_cuda_sm89.cuh:90)
CTA_Q, CTA_K = 128, 64          # sm89 tile (core.py:790-793: CTA_Q=128, CTA_K=64, WARP_Q=32, WARP_K=64)


def _fma32(a: torch.Tensor, b: torch.Tensor, c) -> torch.Tensor:
    """fp32 fused multiply-add: the product of two fp32 numbers is exact in fp64, so one fp64 addition followed by
    the rounding to fp32 reproduces fmaf up to a double-rounding event of probability ~2^-29."""
    return (a.double() * b.double() + torch.as_tensor(c, dtype=torch.float32).double()).float()


def _exp2_32(x: torch.Tensor) -> torch.Tensor:
    return torch.exp2(x.double()).float()


def _e4m3_satfinite(x: torch.Tensor) -> torch.Tensor:
    """cvt.rn.satfinite.e4m3x2.f32 (numeric_conversion.cuh:46-61): RNE, saturating at +-448.  torch's cast is RNE
    but maps |x| >= 480 to NaN, hence the clamp."""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


# ------------------------------------------------------------------------------------------------ quantisers
def quant_int8_cuda(x: torch.Tensor, group: torch.Tensor, ngroups: int, mean: torch.Tensor | None = None):
    """QuantInt8Kernel (fused.cu:64-198): fp32 (x - mean), amax floor 1e-7, scale = amax/127,
    q = float_to_int8_rn(x * (127/amax)) (cvt.rni.sat.s8.f32: nearest-even, saturating).
    x [L, D] fp16/bf16; group [L] slot of every row; returns (int8 [L,D], scale [ngroups])."""
    xf = x.float()
    if mean is not None:
        xf = xf - mean.float()[None, :]
    amax = torch.full((ngroups,), 1e-7, dtype=torch.float32)
    amax = amax.scatter_reduce(0, group, xf.abs().amax(dim=1), reduce="amax")
    scale = amax / 127.0
    tmp = (torch.full_like(amax, 127.0) / amax)[group][:, None]     # IEEE division (scalar / tensor is reciprocal * scalar in torch)
    q = torch.round(xf * tmp).clamp(-128, 127).to(torch.int8)      # torch.round = half-to-even
    return q, scale


def quant_int8_per_thread(x: torch.Tensor, group: torch.Tensor, ngroups: int, mean: torch.Tensor | None = None):
    """quant_per_thread.py:39-46 / :85-95: scale = amax/127 + 1e-7; x/scale; += 0.5*sign; truncate to int8.
    A K mean is subtracted beforehand in the input dtype (core.py:775 `k - km`)."""
    if mean is not None:
        x = x - mean[None, :]              # in the input dtype, as torch does in core.py
    xf = x.float()
    amax = torch.zeros(ngroups, dtype=torch.float32).scatter_reduce(0, group, xf.abs().amax(dim=1), reduce="amax")
    scale = amax / 127.0 + 1e-7
    t = xf / scale[group][:, None]
    t = t + 0.5 * torch.where(t >= 0, 1.0, -1.0)
    return t.to(torch.int8), scale          # .to(int8) truncates toward zero


def groups(L: int, gran: str, which: str):
    r = torch.arange(L)
    if gran == "per_warp":                   # quant.py:169-178: WARPQ=32 rows of each 128 block; K per 64 block
        return (r // 32, (L + 127) // 128 * 4) if which == "q" else (r // 64, (L + 63) // 64)
    if gran == "per_thread":                 # quant_per_thread.py:27-37 (Q), :75-83 (K)
        if which == "q":
            return (r // 32) * 8 + r % 8, (L + 127) // 128 * 32
        return (r // 64) * 4 + (r % 8) // 2, (L + 63) // 64 * 4
    raise ValueError(gran)


def quant_v_fp8(v: torch.Tensor, smooth_v: bool, scale_max: float = 448.0):
    """per_channel_fp8 (quant.py:224-293): transpose + zero-pad (fused.cu:283-286), then MeanScaleKernel: statistics over
    the ceil16(L) tokens of the padded array, mean = sum / ceil16(L), amax = max(|max - mean|, |min - mean|),
    scale = amax / scale_max, bytes = e4m3((v - mean) * (scale_max / amax)).  v [L, D]; returns logical-layout bytes [L, D]."""
    L, D = v.shape
    L16 = (L + 15) // 16 * 16
    vp = torch.zeros(L16, D, dtype=torch.float32)
    vp[:L] = v.float()
    mx, mn = vp.amax(dim=0), vp.amin(dim=0)
    if smooth_v:
        mean = (vp.double().sum(dim=0) / L16).float()
        amax = torch.maximum((mx - mean).abs(), (mn - mean).abs())
    else:
        mean = None
        amax = torch.maximum(mx.abs(), mn.abs())
    scale = amax / scale_max
    recp = torch.full_like(amax, scale_max) / amax                  # IEEE division, see quant_int8_cuda
    x = v.float()
    if smooth_v:
        x = x - mean[None, :]
    return _e4m3_satfinite(x * recp[None, :]), scale, mean


# ------------------------------------------------------------------------------------------------ attention
def attn_fp8(q8, k8, v8, q_scale, q_slot, k_scale, k_slot, v_scale, v_mean, *, causal: bool, sm_scale: float,
             two_level: bool, out_dtype: torch.dtype):
    """One (batch, head).  q8 [Lq,D] int8, k8 [Lk,D] int8, v8 [Lk,D] float8_e4m3fn; q_scale/k_scale are the per-(b,h)
    slot vectors, q_slot/k_slot the slot of every row.  Returns (o [Lq,D] out_dtype, lse [Lq] fp32 in log2 units)."""
    Lq, D = q8.shape
    Lk = k8.shape[0]
    sm = torch.tensor(sm_scale, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)   # sm_scale *= math::log2e
    vf = v8.float()
    o = torch.empty(Lq, D, dtype=out_dtype)
    lse = torch.empty(Lq, dtype=torch.float32)
    for r0 in range(0, Lq, CTA_Q):
        rows = min(CTA_Q, Lq - r0)
        qi = q8[r0:r0 + rows].double()
        qs = q_scale[q_slot[r0:r0 + rows]]
        m = torch.full((rows,), -5000000.0)                 # sm89.cuh:165
        d = torch.ones(rows)                                # sm89.cuh:166 (the first o_scale = exp2(-5e6 - m) is exactly 0)
        RO = torch.zeros(rows, D)
        kend = min(Lk, r0 + CTA_Q) if causal else Lk        # sm89.cuh:237-241
        for n0 in range(0, kend, CTA_K):
            nk = min(CTA_K, Lk - n0)
            S = (qi @ k8[n0:n0 + nk].double().T).float()    # exact integers (|S| < 2^24)
            dequant = qs[:, None] * k_scale[k_slot[n0:n0 + nk]][None, :]      # q_scale * K_scale   (sm89.cuh:264)
            scale = sm * dequant                                              # original_sm_scale * dequant_scale (:266)
            keep = torch.ones(rows, nk, dtype=torch.bool)
            if causal:                                       # kv_idx > q_idx masked (attn_utils.cuh:308-310)
                keep = (n0 + torch.arange(nk))[None, :] <= (r0 + torch.arange(rows))[:, None]
            m_temp = _fma32(S, scale, -S_FP8_OFFSET)         # fma(max RS, sm_scale, -offset); FMA is monotone in RS
            m_temp = torch.where(keep, m_temp, torch.tensor(-float("inf"))).amax(dim=1)
            m_new = torch.maximum(m, m_temp)
            o_scale = _exp2_32(m - m_new)
            P = _exp2_32(_fma32(S, scale, -m_new[:, None]))
            P = torch.where(keep, P, torch.tensor(0.0))
            rs = torch.zeros(rows)
            for j in range(nk):                              # accumulate_d on the CUDA cores: fp32 sum of the fp32 P
                rs = rs + P[:, j]
            d = d * o_scale + rs
            P8 = _e4m3_satfinite(P).float()
            if two_level:                                    # RO_temp from zero, then RO = RO * o_scale + RO_temp
                T = torch.zeros(rows, D)
                for j in range(nk):
                    T = T + P8[:, j:j + 1] * vf[n0 + j][None, :]
                RO = RO * o_scale[:, None] + T
            else:
                RO = RO * o_scale[:, None]
                for j in range(nk):
                    RO = RO + P8[:, j:j + 1] * vf[n0 + j][None, :]
            m = m_new
        x = RO / d[:, None]                                  # normalize_d
        x = x * v_scale[None, :]                             # fuse_v_scale epilogue (sm89.cuh:575-621)
        if v_mean is not None:
            x = x + v_mean[None, :]
        o[r0:r0 + rows] = x.to(out_dtype)
        lse[r0:r0 + rows] = torch.log2(d) + m                # sm89.cuh:691-703
    return o, lse


def attn_f16(q8, k8, vh, q_scale, q_slot, k_scale, k_slot, *, causal: bool, sm_scale: float, out_dtype: torch.dtype):
    """The FP16-PV CUDA kernels (csrc/qattn/qk_int_sv_f16_cuda_sm80.cu:46-671, accum_f32 form) for one (batch, head): no exp offset
    (`update_mdo<..., exp_offset = false>`, :303), P rounded to fp16 by RS_32_to_16 (:316-317), and -- the point of this restatement --
    the denominator accumulated by the TENSOR CORE from those rounded halves (`accumulate_d<..., kTensorCore>` -> mma::rowsum_f16f16f32,
    :313-320, attn_utils.cuh:529-545; DenominatorAccumUnit = kTensorCore in every instantiation :814,989,1164,1348): an FP32 sum of fp16(P).
    vh [Lk, D] float16.  Returns (o [Lq, D] out_dtype, lse [Lq] fp32 in log2 units)."""
    Lq, D = q8.shape
    Lk = k8.shape[0]
    sm = torch.tensor(sm_scale, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)
    vf = vh.float()
    o = torch.empty(Lq, D, dtype=out_dtype)
    lse = torch.empty(Lq, dtype=torch.float32)
    for r0 in range(0, Lq, CTA_Q):
        rows = min(CTA_Q, Lq - r0)
        qi = q8[r0:r0 + rows].double()
        qs = q_scale[q_slot[r0:r0 + rows]]
        m = torch.full((rows,), -5000000.0)
        d = torch.ones(rows)
        RO = torch.zeros(rows, D)
        kend = min(Lk, r0 + CTA_Q) if causal else Lk
        for n0 in range(0, kend, CTA_K):
            nk = min(CTA_K, Lk - n0)
            S = (qi @ k8[n0:n0 + nk].double().T).float()
            scale = sm * (qs[:, None] * k_scale[k_slot[n0:n0 + nk]][None, :])
            keep = torch.ones(rows, nk, dtype=torch.bool)
            if causal:
                keep = (n0 + torch.arange(nk))[None, :] <= (r0 + torch.arange(rows))[:, None]
            m_temp = torch.where(keep, _fma32(S, scale, 0.0), torch.tensor(-float("inf"))).amax(dim=1)     # max(RS) * sm_scale
            m_new = torch.maximum(m, m_temp)
            o_scale = _exp2_32(m - m_new)
            P = torch.where(keep, _exp2_32(_fma32(S, scale, -m_new[:, None])), torch.tensor(0.0))
            P16 = P.to(torch.float16).float()                # RS_32_to_16
            rs = torch.zeros(rows)
            for j in range(nk):                              # the tensor core's FP32 accumulation of the fp16 P
                rs = rs + P16[:, j]
            d = d * o_scale + rs
            RO = RO * o_scale[:, None]
            for j in range(nk):
                RO = RO + P16[:, j:j + 1] * vf[n0 + j][None, :]
            m = m_new
        o[r0:r0 + rows] = (RO / d[:, None]).to(out_dtype)
        lse[r0:r0 + rows] = torch.log2(d) + m
    return o, lse


def sageattn_fp8_cuda(q, k, v, *, is_causal: bool, qk_quant_gran: str, smooth_k: bool = True, smooth_v: bool = False,
                      pv_accum_dtype: str = "fp32+fp32", sm_scale: float | None = None):
    """sageattn_qk_int8_pv_fp8_cuda (core.py:636-826) on HND CPU tensors [B,H,L,D] (D in {64,128}), restated end to end.
    Returns (o, lse_log2_units, aux) with aux = the quantised intermediates for bit comparisons."""
    B, Hq, Lq, D = q.shape
    Hkv, Lk = k.shape[1], k.shape[2]
    g = Hq // Hkv
    if sm_scale is None:
        sm_scale = D ** -0.5
    km = k.float().mean(dim=2).to(k.dtype) if smooth_k else None            # core.py:773 (fp32 sum, one rounding)
    if smooth_v and pv_accum_dtype != "fp32":
        smooth_v = False                                                      # core.py:797-803
    o = torch.empty_like(q)
    lse = torch.empty(B, Hq, Lq, dtype=torch.float32)
    gq, nq = groups(Lq, qk_quant_gran, "q")
    gk, nk = groups(Lk, qk_quant_gran, "k")
    quant = quant_int8_cuda if qk_quant_gran == "per_warp" else quant_int8_per_thread
    aux = dict(q8=torch.empty(B, Hq, Lq, D, dtype=torch.int8), k8=torch.empty(B, Hkv, Lk, D, dtype=torch.int8),
               qs=torch.empty(B, Hq, nq), ks=torch.empty(B, Hkv, nk), v8=torch.empty(B, Hkv, Lk, D, dtype=torch.uint8),
               vs=torch.empty(B, Hkv, D), vm=torch.empty(B, Hkv, D) if smooth_v else None, km=km)
    for b in range(B):
        kq = []
        for hk in range(Hkv):
            k8, ks = quant(k[b, hk], gk, nk, mean=None if km is None else km[b, hk])
            v8, vs, vm = quant_v_fp8(v[b, hk], smooth_v)
            aux["k8"][b, hk], aux["ks"][b, hk], aux["v8"][b, hk], aux["vs"][b, hk] = k8, ks, v8.view(torch.uint8), vs
            if smooth_v:
                aux["vm"][b, hk] = vm
            kq.append((k8, ks, v8, vs, vm))
        for h in range(Hq):
            q8, qs = quant(q[b, h], gq, nq)
            aux["q8"][b, h], aux["qs"][b, h] = q8, qs
            k8, ks, v8, vs, vm = kq[h // g]
            o[b, h], lse[b, h] = attn_fp8(q8, k8, v8, qs, gq, ks, gk, vs, vm, causal=is_causal, sm_scale=sm_scale,
                                          two_level=(pv_accum_dtype != "fp32"), out_dtype=q.dtype)
    return o, lse, aux
