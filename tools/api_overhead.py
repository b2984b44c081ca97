#!/usr/bin/env python3
"""Host-side cost of one sageattn() call on a tiny problem (the GPU work is ~30 us): wall time per call and the
top entries of a cProfile run.  usage: api_overhead.py"""
import cProfile, os, pstats, sys, time, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa

q, k, v = (torch.randn(1, 8, 256, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for _ in range(20):
    sa.sageattn(q, k, v, is_causal=True)
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n):
    sa.sageattn(q, k, v, is_causal=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e6*(t1-t0)/n:.1f} us/call, wall incl. drain {1e6*(t2-t0)/n:.1f} us/call")
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    sa.sageattn(q, k, v, is_causal=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:30]))
