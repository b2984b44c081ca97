#!/usr/bin/env python3
"""V rows in place (fp16 inputs, FP16 PV) against the V tile image: the whole call and the attention kernel alone, interleaved in one process
(HIP events, medians).  usage: vrows_ab.py [c2|c2l|n2k ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import sageattention_amd as sa
from sageattention_amd import core, quant as sq

dev = torch.device("cuda:0")
SHAPES = {"c2": dict(B=2, H=32, N=4096, D=128, causal=True), "c2nc": dict(B=2, H=32, N=4096, D=128, causal=False),
          "c2l": dict(B=2, H=32, N=16384, D=128, causal=True), "n2k": dict(B=2, H=32, N=2048, D=128, causal=True),
          "d64": dict(B=2, H=48, N=8192, D=64, causal=False)}


def med(fn, reps=24):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        t.append(a.elapsed_time(b) * 1e3)
    return sorted(t)[len(t) // 2]


for name in (sys.argv[1:] or ["c2", "c2nc", "c2l", "n2k", "d64"]):
    c = SHAPES[name]
    g = torch.Generator(device="cpu").manual_seed(3)
    q, k, v = (torch.randn(c["B"], c["H"], c["N"], c["D"], generator=g).half().to(dev) for _ in range(3))
    fl = 4.0 * c["B"] * c["H"] * c["N"] * c["N"] * c["D"] / (2 if c["causal"] else 1)
    sm = c["D"] ** -0.5
    rows = {}
    # whole calls
    for api, fn in (("fp16_cuda", sa.sageattn_qk_int8_pv_fp16_cuda), ("fp16_triton", sa.sageattn_qk_int8_pv_fp16_triton)):
        o = {m: fn(q, k, v, is_causal=c["causal"], v_in_place=m) for m in (False, True)}
        torch.cuda.synchronize()
        assert torch.equal(o[False], o[True]), f"{name} {api}: outputs differ"
        t = {m: [] for m in (False, True)}
        for rnd in range(3):
            for m in (False, True):
                t[m].append(med(lambda: fn(q, k, v, is_causal=c["causal"], v_in_place=m), 12))
        rows[api] = {m: sorted(t[m])[1] for m in t}
    # the attention kernels alone (operands from the call's own pre-pass)
    _, k8, ks, vimg, _, _ = sq.prepass_kv_fp8(k, v, "HND", smooth_k=True, qk_quant_gran="per_thread", v_fp16=True)
    _, k8b, ksb, _, _, _ = sq.prepass_kv_fp8(k, None, "HND", smooth_k=True, qk_quant_gran="per_block_triton", v_fp16=True)
    kern = {
        "kernel fused-Q per-thread": {False: lambda: core._attn_fused_q(q, k8, vimg, None, ks, "HND", c["causal"], core._sm_log2(sm), False),
                                      True: lambda: core._attn_fused_q(q, k8, v, None, ks, "HND", c["causal"], core._sm_log2(sm), False, v_rows=True)},
        "kernel fused-Q per-block": {False: lambda: core._attn_fused_qblock(q, k8b, vimg, ksb, "HND", c["causal"], sm * sq.LOG2E, False),
                                     True: lambda: core._attn_fused_qblock(q, k8b, v, ksb, "HND", c["causal"], sm * sq.LOG2E, False, v_rows=True)},
        "pre-pass K + V image | K only": {False: lambda: sq.prepass_kv_fp8(k, v, "HND", smooth_k=True, qk_quant_gran="per_thread", v_fp16=True),
                                          True: lambda: sq.prepass_kv_fp8(k, None, "HND", smooth_k=True, qk_quant_gran="per_thread", v_fp16=True)},
    }
    for label, fns in kern.items():
        t = {m: [] for m in (False, True)}
        for rnd in range(3):
            for m in (False, True):
                t[m].append(med(fns[m], 12))
        rows[label] = {m: sorted(t[m])[1] for m in t}
    for label, r in rows.items():
        tf = (lambda us: f"{fl / us / 1e6:7.1f} TFLOP/s") if "pre-pass" not in label else (lambda us: "")
        print(f"{name:5s} {label:32s} image {r[False]:8.1f} us {tf(r[False])}   rows in place {r[True]:8.1f} us {tf(r[True])}   ({(r[False] / r[True] - 1) * 100:+5.1f} %)", flush=True)
