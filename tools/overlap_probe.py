#!/usr/bin/env python3
"""Upper bound of overlapping the K / V pre-pass (HBM-bound) with the attention kernel (matrix / VALU-bound) inside one sageattn() call
("first measure the upper bound with two half-calls on two streams").  One call over B = 2 is run as
  full        the product's call
  serial      two half calls (batch element 0, then 1) on one stream           -- what splitting alone costs
  two-stream  the half calls on two streams at once                             -- pre-pass || pre-pass, attention || attention
  pipelined   stream A: pre-pass(0), attention(0); stream B: waits for pre-pass(0), then pre-pass(1), attention(1)
              -- attention(0) || pre-pass(1): the overlap a one-launch design with per-head ready flags could reach
usage: overlap_probe.py [c2|c3] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sageattention_amd import core

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[name]
q, k, v = bench.make_inputs(cfg, dev, 99)
fp8 = cfg["pv"] == "fp8"
causal = cfg["causal"]
sm_log2 = core._sm_log2(cfg["D"] ** -0.5)
halves = [(q[i:i + 1], k[i:i + 1], v[i:i + 1]) for i in range(q.shape[0])]


def prepass(qq, kk, vv):
    return core._prepass_kv(qq, kk, vv, "HND", "per_thread", 64, True, False, False, True, v_fp8=fp8, v_fp16=not fp8)


def attend(qq, pp):
    _, _, k8, ks, vimg, vscale, _ = pp
    return core._attn_fused_q(qq, k8, vimg, vscale, ks, "HND", causal, sm_log2, False)[0]


def full():
    return attend(q, prepass(q, k, v))


def serial():
    return [attend(a, prepass(a, b, c)) for a, b, c in halves]


side = torch.cuda.Stream()


def two_stream():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        o1 = attend(halves[1][0], prepass(*halves[1]))
    o0 = attend(halves[0][0], prepass(*halves[0]))
    cur.wait_stream(side)
    return [o0, o1]


def pipelined():
    cur = torch.cuda.current_stream()
    p0 = prepass(*halves[0])
    ev = torch.cuda.Event()
    ev.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        o1 = attend(halves[1][0], prepass(*halves[1]))
    o0 = attend(halves[0][0], p0)
    cur.wait_stream(side)
    return [o0, o1]


want = full()
torch.cuda.synchronize()
fl = bench.flops(cfg)
for nm, fn in (("full", full), ("serial halves", serial), ("two-stream halves", two_stream), ("pipelined halves", pipelined), ("full", full)):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    got = out if torch.is_tensor(out) else torch.cat(out, 0)
    same = torch.equal(got, want)
    xs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        xs.append(a.elapsed_time(b) * 1e3)
    xs.sort()
    med = xs[len(xs) // 2]
    print(f"{name} {nm:20s} median {med:8.1f} us  best {xs[0]:8.1f} us  {fl / med / 1e6:7.1f} TFLOP/s   equal to the full call: {same}", flush=True)
