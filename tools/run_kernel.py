#!/usr/bin/env python3
"""Launch the attention kernel only (pre-quantised operands) a few times -- the target of PMC passes.
usage: run_kernel.py [config | c4 | c4nc | c2t | c2r] [reps] [folded]      (folded: the opt-in FP8 score variant)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
folded = True if (len(sys.argv) > 3 and sys.argv[3] == "folded") else None
dev = torch.device("cuda:0")
if name in ("c4", "c4nc"):        # BASELINE.json configs[3]: the attention launch of sageattn_varlen on the operands its pre-pass produced
    from sageattention_amd import core
    g = torch.Generator(device="cpu").manual_seed(4)
    total = sum(bench.C4_LENS)
    q = torch.randn(total, 32, 128, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    v = torch.randn(total, 8, 128, generator=g).to(torch.bfloat16).to(dev)
    cu = torch.tensor([0] + list(torch.tensor(bench.C4_LENS).cumsum(0)), dtype=torch.int32, device=dev)
    st = core._varlen_prepare(q, k, v, cu, cu, max(bench.C4_LENS), max(bench.C4_LENS), name == "c4", None, True, {})
    torch.cuda.synchronize()
    for _ in range(reps):
        core._varlen_attend(st)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
if name == "c2t":                  # the Triton-named API at the C2 shape: the attention launch alone (per-block Q quantised in the prologue)
    from sageattention_amd import core, quant as sq
    cfg = bench.CONFIGS["c2"]
    q, k, v = bench.make_inputs(cfg, dev, 1234)
    km_s, k8, ks, vimg, _, _ = sq.prepass_kv_fp8(k, v, "HND", smooth_k=True, qk_quant_gran="per_block_triton", v_fp16=True)
    torch.cuda.synchronize()
    for _ in range(reps):
        core._attn_fused_qblock(q, k8, vimg, ks, "HND", True, cfg["D"] ** -0.5 * sq.LOG2E, False)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
if name == "c2r":                  # C2's default route on fp16 inputs: Q quantised in the prologue, V rows read in place
    from sageattention_amd import core
    cfg = bench.CONFIGS["c2"]
    q, k, v = bench.make_inputs(cfg, dev, 1234)
    q8, qs, k8, ks, vimg, vscale, gran, q_warp, sm_log2 = bench.prequantize(cfg, q, k, v)
    torch.cuda.synchronize()
    for _ in range(reps):
        core._attn_fused_q(q, k8, v, None, ks, "HND", True, sm_log2, False, v_rows=True)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
cfg = bench.CONFIGS[name]
q, k, v = bench.make_inputs(cfg, dev, 1234)
ops = bench.prequantize(cfg, q, k, v)
torch.cuda.synchronize()
for _ in range(reps):
    bench.kernel_only_step(cfg, ops, cfg["D"] ** -0.5, folded=folded)
torch.cuda.synchronize()
print("done")
