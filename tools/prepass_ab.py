#!/usr/bin/env python3
"""A/B of sage_prepass_kv between library builds (tools/build_variants.sh with VARIANT_SRC=sage_prepass.hip), interleaved rounds in one
process, bit-equality of every output against the first build.
usage: prepass_ab.py [--shape B,H,N,D] [--dtype bf16|f16] [--rounds 7] [--reps 10] [--cold] tag1 tag2 ...   ('main' = the in-tree library)
--cold: 512 MB of other traffic between launches (an empty Infinity Cache, as inside a sageattn() call)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sageattention_amd import _cabi, quant as sq

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="2,32,8192,128")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--cold", action="store_true")
ap.add_argument("--v16", action="store_true", help="the fp16 V image (FP16-PV entry points) instead of the FP8 one")
ap.add_argument("tags", nargs="+")
args = ap.parse_args()
libs = {}
for tag in args.tags:
    path = os.path.join(ROOT, "sageattention_amd", "libsage_gfx950.so") if tag == "main" else os.path.join(ROOT, "variants", f"libsage_gfx950_{tag}.so")
    _cabi._lib = None
    _cabi.LIB_PATH = path
    libs[tag] = _cabi.load()
B, H, N, D = map(int, args.shape.split(","))
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
k = (torch.randn(B, H, N, D, device=dev, generator=g) * 1.5 + torch.linspace(-2, 3, D, device=dev)).to(dt)
v = torch.randn(B, H, N, D, device=dev, generator=g).to(dt)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if args.cold else None
fn = lambda: sq.prepass_kv_fp8(k, v, v_fp16=args.v16)
ref = None
for tag in args.tags:
    _cabi._lib = libs[tag]
    out = fn()
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    same = all((a is None and b is None) or torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b)
               for a, b in zip(out, ref))
    print(f"{tag:10s} outputs bit-equal to {args.tags[0]}: {same}")
res = {t: [] for t in args.tags}
for r in range(args.rounds):
    for tag in args.tags:
        _cabi._lib = libs[tag]
        fn(); fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in evs:
            if junk is not None:
                junk.add_(1)
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        res[tag].append(sum(a.elapsed_time(b) for a, b in evs) / args.reps * 1e3)
nbytes = 2 * 3 * k.numel() if not args.v16 else (3 + 4) * k.numel()
for tag, us in res.items():
    us = sorted(us)
    med = us[len(us) // 2]
    print(f"{tag:10s} shape {args.shape} {args.dtype}{' cold' if args.cold else ''}: median {med:7.1f} us = {nbytes / med / 1e3:6.0f} GB/s   best {us[0]:7.1f} us")
