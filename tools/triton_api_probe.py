#!/usr/bin/env python3
"""Whole-call time of the Triton-named API (sageattn_qk_int8_pv_fp16_triton) by route: the reference-shaped kernel sequence (K mean, per-block
INT8 Q and K, fp16 V image, attention) against the default (one-launch K/V pre-pass + Q quantisation in the attention prologue)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sageattention_amd as sa

dev = torch.device("cuda:0")
for (B, H, N, D, causal) in [(2, 32, 4096, 128, True), (1, 4, 512, 64, False), (2, 32, 1024, 128, True), (2, 24, 8192, 64, False)]:
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, H, N, D, generator=g).to(torch.float16).to(dev) for _ in range(3))
    fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
    res = {}
    for name, kw in (("sequence", dict(fused_prepass=False, fuse_q_quant=False)), ("default", {})):
        fn = lambda: sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, is_causal=causal, **kw)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 50
    same = torch.equal(sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, is_causal=causal),
                       sa.sageattn_qk_int8_pv_fp16_triton(q, k, v, is_causal=causal, fused_prepass=False, fuse_q_quant=False))
    print(f"B{B} H{H} N{N} D{D} causal={causal}: sequence {res['sequence'] * 1e6:8.1f} us ({fl / res['sequence'] / 1e12:7.1f} TFLOP/s)   "
          f"default {res['default'] * 1e6:8.1f} us ({fl / res['default'] / 1e12:7.1f} TFLOP/s)   bit-identical={same}")
