#!/usr/bin/env python3
"""A/B of two source trees (this repo and a copy of an older commit under variants/old_tree, built there) on whole-call shapes, one process each.
usage: ab_trees.py <tree-root>   -> prints us per call for the split-KV / cross-attention shapes"""
import sys, time
root = sys.argv[1]
sys.path.insert(0, root)
import torch
import sageattention_amd as sa
dev = torch.device("cuda:0")
SHAPES = [("Lq=128 Lk=32768 B1 H32", 1, 32, 32, 128, 32768, 128, False), ("Lq=1024 Lk=16384 B1 H16", 1, 16, 16, 1024, 16384, 128, False),
          ("B1 H8 N8192 causal", 1, 8, 8, 8192, 8192, 128, True), ("cross Lk=512", 2, 24, 24, 16384, 512, 128, False)]
for name, B, Hq, Hkv, Lq, Lk, D, causal in SHAPES:
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, Hq, Lq, D, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(B, Hkv, Lk, D, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(B, Hkv, Lk, D, generator=g).to(torch.bfloat16).to(dev)
    fn = lambda: sa.sageattn(q, k, v, is_causal=causal)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    best = []
    for r in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / 20)
    print(f"{root[-12:]:12s} {name:28s} median {sorted(best)[2] * 1e3:8.1f} us  best {min(best) * 1e3:8.1f} us")
