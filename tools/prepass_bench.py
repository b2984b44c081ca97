#!/usr/bin/env python3
"""HBM-bound pre-pass kernels at the C3 shape: achieved GB/s of algorithmic bytes (read 2 B/elt, write 1 B/elt).
usage: prepass_bench.py [--lib path/to/libsage_gfx950.so]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sageattention_amd import _cabi
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--shape", default="2,32,8192,128")
args = ap.parse_args()
if args.lib:
    _cabi.LIB_PATH = os.path.abspath(args.lib)
from sageattention_amd import quant as sq

B, H, N, D = map(int, args.shape.split(","))
dev = torch.device("cuda:0")
q, k, v = (torch.randn(B, H, N, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
km = sq.channel_mean(k)
elts = B * H * N * D

def t(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

cases = [
    ("per_thread_int8 (q+k, fused k-mean)", lambda: sq.per_thread_int8(q, k, km), 2 * 3 * elts),
    ("per_block_int8 triton-style (q+k)", lambda: sq.per_block_int8(q, k, km), 2 * 3 * elts),
    ("per_warp_int8 cuda-style (q+k)", lambda: sq.per_warp_int8(q, k, km), 2 * 3 * elts),
    ("channel_mean (k)", lambda: sq.channel_mean(k), 2 * elts),
    ("per_channel_fp8 (v: stats + image)", lambda: sq.per_channel_fp8(v), (2 + 2 + 1) * elts),
    ("prep_v_fp16 (v image)", lambda: sq.prep_v_fp16(v), 4 * elts),
]
def sequence():
    m = sq.channel_mean(k)
    sq._quant(k, m, 64, 64, _cabi.GRAN_PER_THREAD, True, _cabi.QSTYLE_TRITON_THREAD, 1.0, "HND", 4)
    sq.per_channel_fp8(v)
if sq.prepass_fused_ok(k):
    cases += [
        ("K+V pre-pass, 6-launch sequence", sequence, 2 * 3 * elts),
        ("K+V pre-pass, one launch (sage_prepass_kv)", lambda: sq.prepass_kv_fp8(k, v), 2 * 3 * elts),
        ("K half, one launch", lambda: sq.prepass_kv_fp8(k, None), 3 * elts),
    ]
for name, fn, nbytes in cases:
    dt = t(fn)
    print(f"{name:46s} {dt*1e6:8.1f} us  {nbytes/dt/1e9:8.0f} GB/s")
