#!/usr/bin/env python3
"""Launch the K / V pre-pass a few times -- the target of PMC passes (tools/pmc_prepass.sh).
usage: run_prepass.py fused|sequence [B,H,N,D] [reps] [--lib path]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sageattention_amd import _cabi
if "--lib" in sys.argv:
    _cabi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from sageattention_amd import quant as sq

mode = sys.argv[1]
B, H, N, D = map(int, (sys.argv[2] if len(sys.argv) > 2 and "," in sys.argv[2] else "2,32,8192,128").split(","))
reps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 3
dev = torch.device("cuda:0")
k = torch.randn(B, H, N, D, device=dev, dtype=torch.bfloat16)
v = torch.randn(B, H, N, D, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 28, device=dev, dtype=torch.float32)          # 1 GiB: flushes the 256 MB Infinity Cache between reps
torch.cuda.synchronize()
for _ in range(reps):
    big.fill_(1.0)
    if mode == "fused":
        sq.prepass_kv_fp8(k, v)
    else:
        m = sq.channel_mean(k)
        sq._quant(k, m, 64, 64, _cabi.GRAN_PER_THREAD, True, _cabi.QSTYLE_TRITON_THREAD, 1.0, "HND", 4)
        sq.per_channel_fp8(v)
torch.cuda.synchronize()
print("done")
