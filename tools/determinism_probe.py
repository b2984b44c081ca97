#!/usr/bin/env python3
"""Run one entry point repeatedly on the same inputs and count calls whose output differs bitwise from the first one.
usage: determinism_probe.py [reps]"""
import os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
def mk(B, H, N, D, dt):
    return [torch.randn(B, H, N, D, device=dev, dtype=torch.float32, generator=g).to(dt) for _ in range(3)]
cases = [
    ("fp16-PV D128 N4096 causal f16", sa.sageattn_qk_int8_pv_fp16_cuda, mk(2, 32, 4096, 128, torch.float16), dict(is_causal=True)),
    ("fp16-PV D128 N4096 non-causal bf16", sa.sageattn_qk_int8_pv_fp16_cuda, mk(2, 32, 4096, 128, torch.bfloat16), dict(is_causal=False)),
    ("fp16-PV D64 N4096 causal f16", sa.sageattn_qk_int8_pv_fp16_cuda, mk(4, 32, 4096, 64, torch.float16), dict(is_causal=True)),
    ("fp8-PV D128 N8192 causal bf16", sa.sageattn, mk(2, 32, 8192, 128, torch.bfloat16), dict(is_causal=True)),
]
for name, fn, (q, k, v), kw in cases:
    for extra in (dict(), dict(fuse_q_quant=False), dict(fused_prepass=False), dict(fuse_q_quant=False, fused_prepass=False), dict(smooth_k=False)):
        if fn is sa.sageattn and ("fuse_q_quant" in extra or "smooth_k" in extra):
            continue
        ref = fn(q, k, v, **kw, **extra)
        bad = 0
        worst = 0.0
        for _ in range(reps):
            o = fn(q, k, v, **kw, **extra)
            if not torch.equal(o, ref):
                bad += 1
                worst = max(worst, float((o.float() - ref.float()).abs().max()))
        print(f"{name:38s} {str(extra):55s} differing calls {bad:3d} / {reps}   max|diff| {worst:.3e}")
