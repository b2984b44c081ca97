#!/usr/bin/env python3
"""kernel-only latency at small N (fixed per-launch / per-workgroup overhead probe)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
for causal in (True, False):
    for n in (64, 128, 256, 512, 1024, 2048):
        cfg = dict(bench.CONFIGS["c3"], N=n, causal=causal)
        q, k, v = bench.make_inputs(cfg, dev, n); ops = bench.prequantize(cfg, q, k, v); sm = 128 ** -0.5
        for _ in range(5): bench.kernel_only_step(cfg, ops, sm)
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): bench.kernel_only_step(cfg, ops, sm)
            b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 20)
        print(f"causal={causal} N={n:5d}: {best*1e3:7.1f} us  {bench.flops(cfg)/best/1e9:7.1f} TF  WGs={2*32*((n+127)//128)}")
