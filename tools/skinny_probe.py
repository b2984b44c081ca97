#!/usr/bin/env python3
"""Lq=128 / Lk=32768 (B=1 H=32 D=128, non-causal): where does the call's time go?  HIP-event time of each stage."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa
from sageattention_amd import core, quant as sq
dev = torch.device("cuda:0")
B, H, Lq, Lk, D = 1, 32, 128, 32768, 128
q = torch.randn(B, H, Lq, D, device=dev, dtype=torch.bfloat16)
k = torch.randn(B, H, Lk, D, device=dev, dtype=torch.bfloat16)
v = torch.randn(B, H, Lk, D, device=dev, dtype=torch.bfloat16)
def t(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
km, k8, ks, vimg, vs, vm = sq.prepass_kv_fp8(k, v)
sm = core._sm_log2(D ** -0.5)
print(f"whole call                 {t(lambda: sa.sageattn(q, k, v)):8.1f} us")
print(f"pre-pass (one launch)      {t(lambda: sq.prepass_kv_fp8(k, v)):8.1f} us")
for S in (0, 4, 8, 16, 32, 64):
    if S == 0:
        us = t(lambda: core._attn_fused_q(q, k8, vimg, vs, ks, "HND", False, sm, False))
    else:
        us = t(lambda: core._attn_fused_q_split(q, k8, vimg, vs, ks, "HND", False, sm, S, False))
    print(f"attention, split_kv={S:<3d}    {us:8.1f} us   ({4.0*B*H*Lq*Lk*D/us/1e6:7.1f} TFLOP/s)")
