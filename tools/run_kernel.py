#!/usr/bin/env python3
"""Launch the attention kernel only (pre-quantised operands) a few times -- the target of PMC passes.
usage: run_kernel.py [config] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
q, k, v = bench.make_inputs(cfg, dev, 1234)
ops = bench.prequantize(cfg, q, k, v)
torch.cuda.synchronize()
for _ in range(reps):
    bench.kernel_only_step(cfg, ops, cfg["D"] ** -0.5)
torch.cuda.synchronize()
print("done")
