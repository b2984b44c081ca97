#!/usr/bin/env python3
"""A/B of library builds on ONE kernel target of tools/run_kernel.py's kind, interleaved in one process (HIP events, medians).
usage: lib_ab.py <c2t|c4|c4nc|c3|c2|e2e:c5|...> tagA tagB ...      ('main' = the in-tree library, other tags = variants/libsage_gfx950_<tag>.so)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sageattention_amd import _cabi, core, quant as sq

name, tags = sys.argv[1], sys.argv[2:]
dev = torch.device("cuda:0")
libs = {}
for tag in tags:
    _cabi._lib = None
    _cabi.LIB_PATH = os.path.join(ROOT, "sageattention_amd", "libsage_gfx950.so") if tag == "main" else os.path.join(ROOT, "variants", f"libsage_gfx950_{tag}.so")
    libs[tag] = _cabi.load()
_cabi._lib = libs[tags[0]]
if name in ("c4", "c4nc"):
    g = torch.Generator(device="cpu").manual_seed(4)
    total = sum(bench.C4_LENS)
    q = torch.randn(total, 32, 128, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    cu = torch.tensor([0] + list(torch.tensor(bench.C4_LENS).cumsum(0)), dtype=torch.int32, device=dev)
    st = core._varlen_prepare(q, k, v, cu, cu, max(bench.C4_LENS), max(bench.C4_LENS), name == "c4", None, True, {})
    step = lambda: core._varlen_attend(st)
    fl = sum(4.0 * 32 * L * L * 128 for L in bench.C4_LENS) / (2 if name == "c4" else 1)
elif name == "c2t":
    cfg = bench.CONFIGS["c2"]
    q, k, v = bench.make_inputs(cfg, dev, 1234)
    km_s, k8, ks, vimg, _, _ = sq.prepass_kv_fp8(k, v, "HND", smooth_k=True, qk_quant_gran="per_block_triton", v_fp16=True)
    step = lambda: core._attn_fused_qblock(q, k8, vimg, ks, "HND", True, cfg["D"] ** -0.5 * sq.LOG2E, False)[0]
    fl = bench.flops(cfg)
elif name.startswith("e2e:"):             # the whole call (default route: fused pre-pass + fused-Q attention)
    cfg = bench.CONFIGS[name[4:]]
    q, k, v = bench.make_inputs(cfg, dev, 1234)
    step = lambda: bench.e2e_step(cfg, q, k, v)
    fl = bench.flops(cfg)
else:
    cfg = bench.CONFIGS[name]
    q, k, v = bench.make_inputs(cfg, dev, 1234)
    opsq = bench.prequantize(cfg, q, k, v)
    step = lambda: bench.kernel_only_step(cfg, opsq, cfg["D"] ** -0.5)
    fl = bench.flops(cfg)
ref = None
t = {tag: [] for tag in tags}
for rnd in range(int(os.environ.get("SAGE_AB_ROUNDS", "4"))):      # (more rounds: differences below 1 %)
    for tag in tags:
        _cabi._lib = libs[tag]
        for _ in range(3):
            o = step()
        torch.cuda.synchronize()
        if ref is None:
            ref = o.clone()
        if not torch.equal(o, ref):          # (SAGE_AB_ALLOW_DIFF=1: variants that are not meant to be bit-equal -- say by how much)
            assert os.environ.get("SAGE_AB_ALLOW_DIFF") == "1", f"{tag}: output differs from {tags[0]}"
            if rnd == 0:
                d = (o.float() - ref.float()).abs().max().item()
                print(f"{name} {tag}: max|o - o_{tags[0]}| = {d:.3e} at max|o| {ref.float().abs().max().item():.3e}", flush=True)
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); step(); b.record(); b.synchronize()
            t[tag].append(a.elapsed_time(b) * 1e3)
for tag in tags:
    xs = sorted(t[tag])
    med = xs[len(xs) // 2]
    print(f"{name} {tag:12s} median {med:9.1f} us  best {xs[0]:9.1f} us  {fl / med / 1e6:7.1f} TFLOP/s", flush=True)
