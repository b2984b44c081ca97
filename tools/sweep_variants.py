#!/usr/bin/env python3
"""kernel-only TFLOPS of several library variants over N (hd128 causal fp8): tools/sweep_variants.py tag..."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from sageattention_amd import _cabi
tags = sys.argv[1:]
libs = {}
for tag in tags:
    path = os.path.join(ROOT, "sageattention_amd", "libsage_gfx950.so") if tag == "main" else os.path.join(ROOT, "variants", f"libsage_gfx950_{tag}.so")
    _cabi._lib = None; _cabi.LIB_PATH = path; libs[tag] = _cabi.load()
dev = torch.device("cuda:0")
for n in (1024, 2048, 4096, 8192, 16384):
    cfg = dict(bench.CONFIGS["c3"], N=n)
    _cabi._lib = libs[tags[0]]
    q, k, v = bench.make_inputs(cfg, dev, n); ops = bench.prequantize(cfg, q, k, v); sm = 128 ** -0.5
    line = f"N={n:6d}"
    for tag in tags:
        _cabi._lib = libs[tag]
        for _ in range(3): bench.kernel_only_step(cfg, ops, sm)
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): bench.kernel_only_step(cfg, ops, sm)
            b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 10)
        line += f"  {tag}: {bench.flops(cfg) / best / 1e9:7.1f} TF ({best*1e3:7.1f} us)"
    print(line)
