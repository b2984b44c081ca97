#!/usr/bin/env python3
"""Persistent launch off / on (ops._PERSISTENT) for non-causal kernel-only launches of C3's shape at N = 8k .. 32k and of C5, interleaved in one
process, HIP events, median of `reps` launches.  usage: persist_ab.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes

import bench
from sageattention_amd import _cabi, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
lib = _cabi.load()
probe = ctypes.c_int32(-1)
ops.launch_hooks(grid_probe=probe).__enter__()      # (for the life of this script)
cases = [("c3 shape, non-causal, N=%d" % n, dict(bench.CONFIGS["c3nc"], N=n)) for n in (8192, 16384, 32768)] + [("c5", bench.CONFIGS["c5"])]
for name, cfg in cases:
    q, k, v = bench.make_inputs(cfg, dev, 7)
    opsq = bench.prequantize(cfg, q, k, v)
    step = lambda: bench.kernel_only_step(cfg, opsq, cfg["D"] ** -0.5)
    t = {False: [], True: []}
    grid = {}
    for rnd in range(2):
        for on in (False, True):
            ops._PERSISTENT = on
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            grid[on] = int(probe.value)
            for _ in range(reps // 2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); step(); b.record(); b.synchronize()
                t[on].append(a.elapsed_time(b) * 1e3)
    med = {on: sorted(x)[len(x) // 2] for on, x in t.items()}
    fl = bench.flops(cfg)
    print(f"{name:34s} items {grid[False]:6d} -> workgroups {grid[True]:6d}   {med[False]:9.1f} us ({fl / med[False] / 1e6:7.1f} TFLOP/s) -> "
          f"{med[True]:9.1f} us ({fl / med[True] / 1e6:7.1f} TFLOP/s)   {100 * (med[False] / med[True] - 1):+.1f} %", flush=True)
    del q, k, v, opsq
    torch.cuda.empty_cache()
ops._PERSISTENT = True
