#!/usr/bin/env python3
"""Whole-call sageattn() throughput over a list of shapes (cross-attention, low batch*heads, short sequences).
usage: shape_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa

SHAPES = [  # name, B, Hq, Hkv, Lq, Lk, D, causal
    ("self-attn C3", 2, 32, 32, 8192, 8192, 128, True),
    ("cross-attn Lk=512 (video x text)", 2, 24, 24, 16384, 512, 128, False),
    ("cross-attn Lk=77", 2, 24, 24, 16384, 77, 128, False),
    ("B=1 H=8 N=8192 causal", 1, 8, 8, 8192, 8192, 128, True),
    ("B=1 H=8 N=8192 causal, split off", 1, 8, 8, 8192, 8192, 128, True),
    ("B=1 H=4 N=16384 causal", 1, 4, 4, 16384, 16384, 128, True),
    ("B=1 H=4 N=16384 causal, split off", 1, 4, 4, 16384, 16384, 128, True),
    ("B=1 H=8 N=32768 causal", 1, 8, 8, 32768, 32768, 128, True),
    ("B=16 H=32 N=1024 causal", 16, 32, 32, 1024, 1024, 128, True),
    ("B=64 H=16 N=256 non-causal D=64", 64, 16, 16, 256, 256, 64, False),
    ("GQA 32/4 N=4096 causal", 2, 32, 4, 4096, 4096, 128, True),
    ("short q, long k: Lq=128 Lk=32768", 1, 32, 32, 128, 32768, 128, False),
    ("short q, long k, split off", 1, 32, 32, 128, 32768, 128, False),
    ("cross-attn B1 H16 Lq=1024 Lk=16384", 1, 16, 16, 1024, 16384, 128, False),
    ("cross-attn B1 H16 Lq=1024 Lk=16384, split off", 1, 16, 16, 1024, 16384, 128, False),
]
dev = torch.device("cuda:0")
for name, B, Hq, Hkv, Lq, Lk, D, causal in SHAPES:
    q = torch.randn(B, Hq, Lq, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, Hkv, Lk, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, Hkv, Lk, D, device=dev, dtype=torch.bfloat16)
    fn = (lambda: sa.sageattn(q, k, v, is_causal=causal, split_kv=0)) if "split off" in name else (lambda: sa.sageattn(q, k, v, is_causal=causal, split_kv="auto"))
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    fl = 4.0 * B * Hq * Lq * Lk * D / (2 if causal else 1)
    print(f"{name:40s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s")
