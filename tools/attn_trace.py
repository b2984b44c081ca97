#!/usr/bin/env python3
"""Per-workgroup phase times of ONE attention launch, from the 100 MHz stamps of a -DSAGE_ATTN_TRACE=1 build of the attention units
(tools/build_variants.sh atrace:"-DSAGE_ATTN_TRACE=1"; run with SAGE_GFX950_LIB=variants/libsage_gfx950_atrace.so).  The stamps land in a
caller-owned buffer handed over as a launch attribute (SageLaunchAttr.trace; an ordinary build ignores it and the tool sees only zeros).
usage: attn_trace.py [c3|c2|c5|n1k|n2k|n4k|c4|c4nc]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sageattention_amd import _cabi, core, ops as sa_ops

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
lib = _cabi.load()
if name in ("c4", "c4nc"):
    g = torch.Generator(device="cpu").manual_seed(4)
    total = sum(bench.C4_LENS)
    q = torch.randn(total, 32, 128, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    cu = torch.tensor([0] + list(torch.tensor(bench.C4_LENS).cumsum(0)), dtype=torch.int32, device=dev)
    st = core._varlen_prepare(q, k, v, cu, cu, max(bench.C4_LENS), max(bench.C4_LENS), name == "c4", None, True, {})
    step = lambda: core._varlen_attend(st)
else:
    cfg = bench.CONFIGS[name]
    q, k, v = bench.make_inputs(cfg, dev, 1234)
    ops = bench.prequantize(cfg, q, k, v)
    step = lambda: bench.kernel_only_step(cfg, ops, cfg["D"] ** -0.5)
for _ in range(5):
    step()
torch.cuda.synchronize()
NW = 1 << 15
tbuf = torch.zeros(16 * NW, dtype=torch.int32, device=dev)
with sa_ops.launch_hooks(trace_buf=tbuf):
    step()
    torch.cuda.synchronize()
buf = tbuf.cpu().numpy().view(np.uint32)
if not buf.any():
    sys.exit("this library has no trace: build it with -DSAGE_ATTN_TRACE=1 and point SAGE_GFX950_LIB at it")
t = buf.reshape(NW, 16)
t = t[t[:, 7] != 0]                                   # workgroups that ran to the end (the ones that exit early leave zeros)
st_ = t[:, :8].astype(np.int64)
st_ = (st_ - st_[:, :1].min()) & 0xffffffff           # relative to the first workgroup's entry
us = st_ / 100.0
life = us[:, 7] - us[:, 0]
span = us[:, 7].max() - us[:, 0].min()
print(f"{name}: {len(t)} workgroups, kernel span {span:.1f} us, mean lifetime {life.mean():.2f} us")
names = ["index chain / geometry", "first DMA issued + Q ready", "wait for the first tile", "key loop", "epilogue barrier", "normalise + transpose + store issue",
         "stores acknowledged"]
tot = life.sum()
for i, nm in enumerate(names):
    d = us[:, i + 1] - us[:, i]
    print(f"   {nm:38s} mean {d.mean():8.2f}  p50 {np.percentile(d, 50):8.2f}  p90 {np.percentile(d, 90):8.2f}  max {d.max():8.2f} us   {100 * d.sum() / tot:5.1f} % of the slot time")
outside = life - (us[:, 4] - us[:, 3])
print(f"   outside the key loop: mean {outside.mean():.2f} us per workgroup = {100 * outside.sum() / tot:.1f} % of the slot time")
# slot idle time per CU: two slots per CU; CU identity = (xcc, se, sh, cu) from HW_ID / XCC_ID
hw, xcc = t[:, 9].astype(np.int64), t[:, 10].astype(np.int64) & 0xf
cu_key = (xcc << 16) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 0xf)
keys = np.unique(cu_key)
busy = np.array([life[cu_key == c].sum() for c in keys])
first = np.array([us[cu_key == c, 0].min() for c in keys])
last = np.array([us[cu_key == c, 7].max() for c in keys])
print(f"   {len(keys)} compute units seen; slot occupancy (sum of lifetimes / (2 x kernel span)): mean {100 * (busy / (2 * span)).mean():.1f} %, min {100 * (busy / (2 * span)).min():.1f} %")
print(f"   first workgroup entry per CU: mean {first.mean():.2f} us (max {first.max():.2f}); last exit per CU: mean {last.mean():.1f} us, min {last.min():.1f} (span {span:.1f})")
# a CU's two slots: the gap in front of every workgroup entry after the CU's first two (entry - the latest exit on that CU before it: what
# the dispatcher needs to refill a freed slot), and the tail (from the moment the CU holds fewer than two workgroups for good to the kernel's end)
gaps, tails = [], []
kend = us[:, 7].max()
for c in keys:
    m = cu_key == c
    ent, ext = np.sort(us[m, 0]), np.sort(us[m, 7])
    for e in ent[2:]:
        j = np.searchsorted(ext, e, side="right") - 1
        if j >= 0:
            gaps.append(e - ext[j])
    # the last moment two workgroups were resident = the second-latest exit if nothing entered after it
    tails.append(kend - max(ext[-2] if len(ext) > 1 else ext[-1], ent[-1]))
gaps, tails = np.array(gaps if gaps else [0.0]), np.array(tails)
print(f"   refill gap (entry - latest earlier exit on the CU): mean {gaps.mean():.2f}  p50 {np.percentile(gaps, 50):.2f}  p90 {np.percentile(gaps, 90):.2f}  max {gaps.max():.2f} us "
      f"x {len(gaps) / len(keys):.1f} refills per CU = {100 * gaps.sum() / len(keys) / (2 * span):.2f} % of 2 x span")
print(f"   tail per CU (fewer than two workgroups resident until the kernel ends): mean {tails.mean():.1f}  max {tails.max():.1f} us = {100 * tails.mean() / (2 * span):.2f} % of 2 x span")
wgs = np.unique(t[:, 11])
if len(wgs) != len(t):          # persistent launch: several items per workgroup
    per = np.array([np.sum(t[:, 11] == w) for w in wgs])
    print(f"   persistent: {len(wgs)} workgroups took {len(t)} items ({per.min()} .. {per.max()} each); workgroups per CU: "
          f"{np.bincount(np.array([np.sum(np.unique(t[cu_key == c, 11]).size) for c in keys]))}")
if os.environ.get("SAGE_TRACE_DUMP"):
    np.save(os.environ["SAGE_TRACE_DUMP"], t)
