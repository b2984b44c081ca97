"""CPU oracle for the SageAttention hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``sageattention_amd``) never does.  The arithmetic lives
in ``sage_oracle.c`` (plain C, each function cites the reference file:line it restates);
this module is a thin numpy/ctypes front end plus the host glue of the reference's Python
API (``/root/reference/sageattention/core.py:160-331`` dense, ``:334-448`` varlen) restated
so that whole-API results can be compared.

Arrays are numpy; fp16/bf16 tensors travel as ``uint16`` bit patterns with an explicit
``dtype`` code (0 = fp16, 1 = bf16).  Layout is always HND-contiguous ``[B, H, L, D]``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsage_oracle.so")
LOG2E = 1.44269504  # the literal the reference uses (quant_per_block.py:87)

F16, BF16 = 0, 1
STYLE_TRITON, STYLE_CUDA, STYLE_TRITON_THREAD = 0, 1, 2
PV_F16_TRITON, PV_F16_F32ACC, PV_F8_TWO_LEVEL, PV_F8_SINGLE = 0, 1, 2, 3
SCORES_EXACT, SCORES_FOLDED = 0, 1      # FP8 PV: the form of the softmax argument (sage_oracle.c, score_mode)


def build(force: bool = False) -> str:
    """Compile libsage_oracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "sage_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_h2f.restype = ctypes.c_float
        _lib.orc_h2f.argtypes = [ctypes.c_uint16]
        _lib.orc_e4m3_2f.restype = ctypes.c_float
        _lib.orc_e4m3_2f.argtypes = [ctypes.c_uint8]
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle takes contiguous arrays"
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- conversions
def convert(x: np.ndarray, kind: str) -> np.ndarray:
    """fp32 -> {"f16","bf16","e4m3"} bit patterns with the oracle's own RNE converters."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    code = {"f16": 0, "bf16": 1, "e4m3": 2}[kind]
    out = np.empty(x.shape, dtype=np.uint8 if code == 2 else np.uint16)
    lib().orc_convert_array(_p(x), _p(out), ctypes.c_long(x.size), ctypes.c_int(code))
    return out


def to_f32(bits: np.ndarray, dtype: int) -> np.ndarray:
    """fp16/bf16 bit patterns -> float32 (exact)."""
    if dtype == F16:
        return bits.view(np.float16).astype(np.float32)
    return (bits.astype(np.uint32) << 16).view(np.float32)


def e4m3_to_f32(b: np.ndarray) -> np.ndarray:
    lut = np.array([lib().orc_e4m3_2f(i) for i in range(256)], dtype=np.float32)
    return lut[b]


# --------------------------------------------------------------------------- scale groups
def group_index(L: int, gran: str, which: str, BLK: int, WARP: int) -> Tuple[np.ndarray, int]:
    """Scale-slot index of every row and slot count per (b, h).

    gran "per_block": row // BLK                       (quant_per_block.py:29-31)
    gran "per_warp":  Q row // WARP, K row // BLK      (quant.py:169-178)
    gran "per_thread": Q (row//WARP)*8 + row%8; K (row//WARP)*4 + (row%8)//2
                                                        (quant_per_thread.py:27-37,75-83)
    """
    r = np.arange(L, dtype=np.int64)
    nblk = (L + BLK - 1) // BLK
    if gran == "per_block" or (gran == "per_warp" and which == "k"):
        return (r // BLK).astype(np.int32), nblk
    if gran == "per_warp":
        return (r // WARP).astype(np.int32), nblk * (BLK // WARP)
    if gran == "per_thread":
        if which == "q":
            return ((r // WARP) * 8 + r % 8).astype(np.int32), nblk * (BLK // WARP) * 8
        return ((r // WARP) * 4 + (r % 8) // 2).astype(np.int32), nblk * (BLK // WARP) * 4
    raise ValueError(gran)


# --------------------------------------------------------------------------- kernels
def quant_int8(x: np.ndarray, dtype: int, group: np.ndarray, ngroups: int, pre_scale: float = 1.0,
               style: int = STYLE_TRITON, mean: Optional[np.ndarray] = None):
    """x [B,H,L,D] uint16 bits -> (int8 [B,H,L,D], scale fp32 [B,H,ngroups])."""
    B, H, L, D = x.shape
    out = np.empty((B, H, L, D), dtype=np.int8)
    scale = np.empty((B, H, ngroups), dtype=np.float32)
    group = np.ascontiguousarray(group, dtype=np.int32)
    rc = lib().orc_quant_int8(_p(x), int(dtype), _p(mean), _p(out), _p(scale), _p(group), int(ngroups),
                              int(B), int(H), int(L), int(D), ctypes.c_float(float(pre_scale)), int(style))
    assert rc == 0
    return out, scale


def quant_v_fp8(v: np.ndarray, dtype: int, scale_max: float = 448.0, mean: Optional[np.ndarray] = None):
    """v [B,H,L,D] uint16 bits -> (e4m3 bytes [B,H,L,D] logical layout, v_scale [B,H,D]).
    mean (fp32 [B,H,D], optional) = smooth_v: subtracted before quantising."""
    B, H, L, D = v.shape
    out = np.empty((B, H, L, D), dtype=np.uint8)
    vs = np.empty((B, H, D), dtype=np.float32)
    if mean is not None:
        mean = np.ascontiguousarray(mean, dtype=np.float32)
    rc = lib().orc_quant_v_fp8(_p(v), int(dtype), _p(out), _p(vs), _p(mean), int(B), int(H), int(L), int(D),
                               ctypes.c_float(float(scale_max)))
    assert rc == 0
    return out, vs


def v_mean_padded16(v: np.ndarray, dtype: int) -> np.ndarray:
    """The reference's smooth_v mean: sum over tokens / ceil16(L) in fp32 (fused.cu:335,381)."""
    L = v.shape[2]
    return (to_f32(v, dtype).astype(np.float64).sum(axis=2) / float((L + 15) // 16 * 16)).astype(np.float32)


def attn(q8, k8, v, q_scale, q_sidx, k_scale, k_sidx, *, causal: bool, c: float, pv_mode: int,
         out_dtype: int, v_scale=None, v_mean=None, return_lse: bool = False, mask_bool=None, mask_add=None, score_mode: int = SCORES_EXACT,
         tile_keys: int = 64):
    """Fused attention on quantised operands; returns (o bits uint16 [B,Hq,Lq,D], lse|None).  ``score_mode`` (FP8 modes): SCORES_EXACT
    = the reference's formula, SCORES_FOLDED = the reassociation of the gfx950 kernels' opt-in folded variant (sage_oracle.c).
    ``tile_keys``: keys per online-softmax iteration, 64 (sm80 / sm89 / Triton kernels, and every gfx950 kernel) or 128 (the sm90 kernel's CTA_K)."""
    B, Hq, Lq, D = q8.shape
    _, Hkv, Lk, _ = k8.shape
    o = np.empty((B, Hq, Lq, D), dtype=np.uint16)
    lse = np.empty((B, Hq, Lq), dtype=np.float32) if return_lse else None
    q_sidx = np.ascontiguousarray(q_sidx, dtype=np.int32)
    k_sidx = np.ascontiguousarray(k_sidx, dtype=np.int32)
    rc = lib().orc_attn_ex(_p(q8), _p(k8), _p(v), _p(o), _p(lse),
                        _p(q_scale), _p(q_sidx), int(q_scale.shape[-1]),
                        _p(k_scale), _p(k_sidx), int(k_scale.shape[-1]), _p(v_scale),
                        _p(None if v_mean is None else np.ascontiguousarray(v_mean, dtype=np.float32)),
                        _p(None if mask_bool is None else np.ascontiguousarray(np.broadcast_to(mask_bool, (B, Hq, Lq, Lk)), dtype=np.uint8)),
                        _p(None if mask_add is None else np.ascontiguousarray(np.broadcast_to(mask_add, (B, Hq, Lq, Lk)), dtype=np.float32)),
                        int(B), int(Hq), int(Hkv), int(Lq), int(Lk), int(D), int(causal),
                        ctypes.c_float(float(c)), int(pv_mode), int(out_dtype), int(score_mode), int(tile_keys))
    assert rc == 0, "orc_attn_ex rejected the arguments"
    return o, lse


# --------------------------------------------------------------------------- API-level restatement
def _pad_head_dim(x: np.ndarray, dtype: int):
    D = x.shape[-1]
    if D < 64:
        Dp = 64
    elif 64 < D < 128:
        Dp = 128
    elif D > 128:
        raise ValueError(f"Unsupported head_dim: {D}")
    else:
        return x
    pad = np.zeros(x.shape[:-1] + (Dp - D,), dtype=x.dtype)
    return np.ascontiguousarray(np.concatenate([x, pad], axis=-1))


def k_mean(k: np.ndarray, dtype: int) -> np.ndarray:
    """km = k.mean(dim=seq) in the input dtype (core.py:280): fp32 sum, one rounding."""
    kf = to_f32(k, dtype).astype(np.float64).mean(axis=2).astype(np.float32)
    return convert(kf, "f16" if dtype == F16 else "bf16")


def sageattn_dense(q, k, v, dtype: int, *, is_causal=False, sm_scale=None, smooth_k=True,
                   qk_quant_gran="per_block", pv="f16_triton", return_lse=False, km=None, warpq=32,
                   smooth_v=False, vm=None, mask_bool=None, mask_add=None, blkk=64, fp8_scores="exact", single_level=False, tile_keys=64):
    """Whole-API restatement on HND arrays of fp16/bf16 bits.

    pv "f16_triton": sageattn_qk_int8_pv_fp16_triton (core.py:160-331), per-block quant with
        sm_scale*log2e folded into Q.
    pv "f8": sageattn_qk_int8_pv_fp8_cuda (core.py:636-826) with fp32+fp32 two-level
        accumulation; qk_quant_gran per_warp | per_thread (| per_block, our extension).
    pv "f16": sageattn_qk_int8_pv_fp16_cuda pv_accum_dtype="fp32" (core.py:451-633).
    fp8_scores ("exact" | "folded", pv "f8" only): the form of the softmax argument, see ``attn`` (the product's default is "exact").
    tile_keys (pv "f8"): 64, or 128 = the sm90 kernel's own iteration (qk_int_sv_f8_cuda_sm90.cu:285-356), see ``attn``.
    single_level (pv "f8"): pv_accum_dtype="fp32", every tile accumulated straight into the output.
    warpq / blkk: scale-group sizes of the CUDA-named APIs -- WARPQ 32, or 16 for head_dim 128 with
        "fp16+fp32" (core.py:602-604); the sm90 entry point uses WARPQ 16 and BLKK = WARPK = 128 (core.py:964-970).
    Returns (o bits [B,Hq,Lq,D0], lse or None, aux dict of intermediates).
    """
    D0 = q.shape[-1]
    q, k, v = (_pad_head_dim(t, dtype) for t in (q, k, v))
    B, Hq, Lq, D = q.shape
    Hkv, Lk = k.shape[1], k.shape[2]
    if sm_scale is None:
        sm_scale = 1.0 / (D0 ** 0.5)
    if smooth_k and km is None:
        km = k_mean(k, dtype)
    if not smooth_k:
        km = None
    vh = v if dtype == F16 else convert(to_f32(v, dtype), "f16")   # core.py:297-298,613
    if pv == "f16_triton" or qk_quant_gran == "per_block":
        # Triton API: "triton" rounding (quant_per_block.py); per_block under the CUDA-named APIs
        # (a gfx950 extension) uses the CUDA quantiser's rounding (quant.py:22-103 -> fused.cu:64-198)
        style = STYLE_TRITON if pv == "f16_triton" else STYLE_CUDA
        gq, nq = group_index(Lq, "per_block", "q", 128, 128)
        gk, nk = group_index(Lk, "per_block", "k", 64, 64)
        q8, qs = quant_int8(q, dtype, gq, nq, pre_scale=np.float32(sm_scale * LOG2E), style=style)
        k8, ks = quant_int8(k, dtype, gk, nk, style=style, mean=km)
        c = 1.0
    elif qk_quant_gran == "per_warp":
        gq, nq = group_index(Lq, "per_warp", "q", 128, warpq)     # WARPQ 32, or 16 (core.py:602)
        gk, nk = group_index(Lk, "per_warp", "k", blkk, blkk)
        q8, qs = quant_int8(q, dtype, gq, nq, style=STYLE_CUDA)
        k8, ks = quant_int8(k, dtype, gk, nk, style=STYLE_CUDA, mean=km)
        c = float(np.float32(sm_scale) * np.float32(LOG2E))
    elif qk_quant_gran == "per_thread":
        gq, nq = group_index(Lq, "per_thread", "q", 128, warpq)
        gk, nk = group_index(Lk, "per_thread", "k", blkk, blkk)
        q8, qs = quant_int8(q, dtype, gq, nq, style=STYLE_TRITON_THREAD)
        k8, ks = quant_int8(k, dtype, gk, nk, style=STYLE_TRITON_THREAD, mean=km)
        c = float(np.float32(sm_scale) * np.float32(LOG2E))
    else:
        raise ValueError(qk_quant_gran)
    aux = dict(q8=q8, qs=qs, k8=k8, ks=ks, km=km, gq=gq, gk=gk, c=c)
    if pv == "f8":
        # smooth_v (pv_accum_dtype="fp32" only, core.py:797-803): mean subtracted before quantising,
        # added back in the epilogue; vm may be handed in (fp32 [B,Hkv,D]) so checker and kernel share it
        if smooth_v and vm is None:
            vm = v_mean_padded16(v, dtype)
        v8, vs = quant_v_fp8(v, dtype, mean=vm if smooth_v else None)
        aux.update(v8=v8, vs=vs, vm=vm)
        o, lse = attn(q8, k8, v8, qs, gq, ks, gk, causal=is_causal, c=c, pv_mode=PV_F8_SINGLE if single_level else PV_F8_TWO_LEVEL,
                      out_dtype=dtype, v_scale=vs, v_mean=vm if smooth_v else None, return_lse=return_lse,
                      score_mode={"exact": SCORES_EXACT, "folded": SCORES_FOLDED}[fp8_scores], tile_keys=tile_keys)
    else:
        mode = PV_F16_TRITON if pv == "f16_triton" else PV_F16_F32ACC
        if smooth_v:   # sub_mean (quant.py:182-222): vm = v.mean(seq) in the input dtype, (v - vm) -> fp16
            if vm is None:
                vm = to_f32(k_mean(v, dtype), dtype)
            vh = convert(to_f32(v, dtype) - vm[:, :, None, :], "f16")
            aux.update(vm=vm)
        o, lse = attn(q8, k8, vh, qs, gq, ks, gk, causal=is_causal, c=c, pv_mode=mode,
                      out_dtype=dtype, v_mean=vm if smooth_v else None, return_lse=return_lse,
                      mask_bool=mask_bool, mask_add=mask_add)
    o = np.ascontiguousarray(o[..., :D0])
    if return_lse:
        lse = lse / np.float32(LOG2E)
        if smooth_k:   # core.py:289-293,328-329
            g = Hq // Hkv
            kmf = np.repeat(to_f32(km, dtype), g, axis=1)                  # [B,Hq,D]
            corr = np.einsum("bhld,bhd->bhl", to_f32(q, dtype), kmf)
            corr = to_f32(convert(corr, "f16" if dtype == F16 else "bf16"), dtype)  # matmul result in input dtype
            lse = lse + corr * np.float32(sm_scale)
    return o, lse, aux


def sageattn_varlen(q, k, v, dtype: int, cu_q, cu_k, *, is_causal=False, sm_scale=None, smooth_k=True, km=None):
    """sageattn_varlen (core.py:334-448) on packed [sum L, H, D] arrays of fp16/bf16 bits.
    km (bits [1, Hkv, D], optional): the K mean to use instead of computing it (host plumbing, as in sageattn_dense)."""
    D0 = q.shape[-1]
    q, k, v = (_pad_head_dim(t, dtype) for t in (q, k, v))
    if sm_scale is None:
        sm_scale = 1.0 / (D0 ** 0.5)
    vh = v if dtype == F16 else convert(to_f32(v, dtype), "f16")
    if smooth_k:   # mean over ALL packed tokens, then k - km in the input dtype (core.py:432-434)
        kf = to_f32(k, dtype)
        if km is None:
            km = convert(kf.astype(np.float64).mean(axis=0, keepdims=True).astype(np.float32),
                         "f16" if dtype == F16 else "bf16")
        km = to_f32(np.asarray(km).reshape(1, k.shape[1], k.shape[2]), dtype)
        k = convert(kf - km, "f16" if dtype == F16 else "bf16")
    o = np.zeros(q.shape, dtype=np.uint16)
    for b in range(len(cu_q) - 1):
        qs_, qe = int(cu_q[b]), int(cu_q[b + 1])
        ks_, ke = int(cu_k[b]), int(cu_k[b + 1])
        if qe == qs_:
            continue
        qb = np.ascontiguousarray(q[qs_:qe].transpose(1, 0, 2))[None]
        kb = np.ascontiguousarray(k[ks_:ke].transpose(1, 0, 2))[None]
        vb = np.ascontiguousarray(vh[ks_:ke].transpose(1, 0, 2))[None]
        gq, nq = group_index(qe - qs_, "per_block", "q", 128, 128)
        gk, nk = group_index(ke - ks_, "per_block", "k", 64, 64)
        q8, qsc = quant_int8(qb, dtype, gq, nq, pre_scale=np.float32(sm_scale * LOG2E))
        k8, ksc = quant_int8(kb, dtype, gk, nk)
        ob, _ = attn(q8, k8, vb, qsc, gq, ksc, gk, causal=is_causal, c=1.0, pv_mode=PV_F16_TRITON,
                     out_dtype=dtype)
        o[qs_:qe] = ob[0].transpose(1, 0, 2)
    return np.ascontiguousarray(o[..., :D0])
