#!/usr/bin/env python3
"""A/B kernel variants (built by tools/build_variants.sh) on the same pre-quantised operands, interleaved rounds.
usage: variant_bench.py [--config c3] [--rounds 5] [--reps 5] tag1 tag2 ...   ('main' = the in-tree library)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sageattention_amd import _cabi

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("tags", nargs="+")
args = ap.parse_args()

libs = {}
for tag in args.tags:
    path = os.path.join(ROOT, "sageattention_amd", "libsage_gfx950.so") if tag == "main" else os.path.join(ROOT, "variants", f"libsage_gfx950_{tag}.so")
    _cabi._lib = None
    _cabi.LIB_PATH = path
    libs[tag] = _cabi.load()

cfg = bench.CONFIGS[args.config]
dev = torch.device("cuda:0")
q, k, v = bench.make_inputs(cfg, dev, 1234)
_cabi._lib = libs[args.tags[0]]
ops = bench.prequantize(cfg, q, k, v)
sm = cfg["D"] ** -0.5
fl = bench.flops(cfg)
ref = None
res = {t: [] for t in args.tags}
for tag in args.tags:                      # correctness vs the first variant
    _cabi._lib = libs[tag]
    o = bench.kernel_only_step(cfg, ops, sm)
    torch.cuda.synchronize()
    if ref is None:
        ref = o.float()
    d = (o.float() - ref).abs().max().item()
    print(f"{tag:12s} max|o - o_{args.tags[0]}| = {d:.3e} (max|o| {ref.abs().max().item():.3e}) finite={bool(torch.isfinite(o.float()).all())}")
for r in range(args.rounds):
    for tag in args.tags:
        _cabi._lib = libs[tag]
        for _ in range(2):
            bench.kernel_only_step(cfg, ops, sm)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.reps):
            bench.kernel_only_step(cfg, ops, sm)
        b.record()
        torch.cuda.synchronize()
        res[tag].append(a.elapsed_time(b) / args.reps)
out = {}
for tag, ms in res.items():
    ms = sorted(ms)
    med = ms[len(ms) // 2]
    out[tag] = dict(ms_med=round(med, 4), ms_min=round(ms[0], 4), tflops_med=round(fl / med / 1e9, 1), tflops_best=round(fl / ms[0] / 1e9, 1))
    print(f"{tag:12s} median {med:.4f} ms = {fl / med / 1e9:7.1f} TFLOPS   best {ms[0]:.4f} ms = {fl / ms[0] / 1e9:7.1f}")
print(json.dumps({"config": args.config, "variants": out}))
