#!/usr/bin/env python3
"""FP16-PV pre-pass at a bench shape: the one launch (K mean + INT8 K + fp16 V image) against its two halves -- the K half (one launch, v=None)
and the V image (prep_v_fp16, a pure permutation: no statistics, no head barrier) -- in series on one stream and side by side on two.

    python tools/kv_split_probe.py [c2|c3|n1k ..]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sageattention_amd import quant as sq

SHAPES = {"c2": (2, 32, 4096, 128), "c3": (2, 32, 8192, 128), "n1k": (2, 32, 1024, 128), "n16k": (2, 32, 16384, 128), "c5": (2, 48, 17776, 64)}
DEV = torch.device("cuda:0")


def timed(fn, reps=60, warm=15):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


for name in (sys.argv[1:] or ["c2"]):
    B, H, L, D = SHAPES[name]
    g = torch.Generator(device="cuda").manual_seed(1)
    k = (torch.randn(B, H, L, D, device=DEV, generator=g) + 1.0).half()
    v = torch.randn(B, H, L, D, device=DEV, generator=g).half()
    side = torch.cuda.Stream()
    e_fork, e_join = torch.cuda.Event(), torch.cuda.Event()

    def one():
        return sq.prepass_kv_fp8(k, v, "HND", qk_quant_gran="per_block_triton", v_fp16=True)

    def series():
        a = sq.prepass_kv_fp8(k, None, "HND", qk_quant_gran="per_block_triton")
        return a, sq.prep_v_fp16(v, "HND")

    def two_streams():
        e_fork.record()
        with torch.cuda.stream(side):
            side.wait_event(e_fork)
            img = sq.prep_v_fp16(v, "HND")
            e_join.record()
        a = sq.prepass_kv_fp8(k, None, "HND", qk_quant_gran="per_block_triton")
        torch.cuda.current_stream().wait_event(e_join)
        return a, img

    def k_only():
        return sq.prepass_kv_fp8(k, None, "HND", qk_quant_gran="per_block_triton")

    def v_only():
        return sq.prep_v_fp16(v, "HND")

    ref = one()
    a, img = two_streams()
    torch.cuda.synchronize()
    same = torch.equal(ref[1], a[1]) and torch.equal(ref[2], a[2]) and torch.equal(ref[3], img)
    byt = B * H * L * D * 7
    print(f"== {name}: B{B} H{H} L{L} D{D}; outputs of the split equal the one launch: {same}")
    for label, fn, nb in (("one launch (K + V)", one, byt), ("K half, then V image (one stream)", series, byt), ("K half || V image (two streams)", two_streams, byt),
                          ("K half alone", k_only, B * H * L * D * 3), ("V image alone", v_only, B * H * L * D * 4)):
        med, best = timed(fn)
        print(f"   {label:38s} median {med:7.1f} us  best {best:7.1f} us   {nb / med / 1e6:6.2f} TB/s at the median")
