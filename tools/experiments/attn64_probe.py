#!/usr/bin/env python3
"""A/B of the two FP8-PV D=128 attention kernels (sage_set_attn64_mode 0 / 1): agreement of their outputs, accuracy against
fp32 SDPA, and kernel-only / whole-call timings over the BASELINE sweep.  GPU only.

    python tools/attn64_probe.py [--quick] [--sweep]
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa  # noqa: E402
from sageattention_amd import _cabi, core, quant as sq  # noqa: E402

DEV = torch.device("cuda:0")
lib = _cabi.load()


def sdpa32(q, k, v, causal):
    g = q.shape[1] // k.shape[1]
    kk = k.float().repeat_interleave(g, 1)
    vv = v.float().repeat_interleave(g, 1)
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), kk) / math.sqrt(q.shape[-1])
    if causal:
        Lq, Lk = q.shape[2], k.shape[2]
        m = torch.arange(Lk, device=q.device)[None, :] > torch.arange(Lq, device=q.device)[:, None]
        s = s.masked_fill(m, float("-inf"))
    return torch.softmax(s, -1) @ vv


def run(mode, fn):
    lib.sage_set_attn64_mode(mode)
    out = fn()
    torch.cuda.synchronize()
    return out


def agree(tag, fn, ref=None):
    o0 = run(0, fn)
    o1 = run(1, fn)
    lse0 = lse1 = None
    if isinstance(o0, tuple):
        o0, lse0 = o0
        o1, lse1 = o1
    a, b = o0.float(), o1.float()
    scale = a.abs().max().item()
    err = (a - b).abs().max().item()
    rec = dict(case=tag, max_o=scale, max_diff_64_vs_128=err, finite=bool(torch.isfinite(b).all().item()))
    if lse0 is not None:
        rec["lse_diff"] = (lse0 - lse1).abs().max().item()
    if ref is not None:
        r = ref.float()
        rec["cos_128"] = torch.nn.functional.cosine_similarity(a.flatten(), r.flatten(), dim=0).item()
        rec["cos_64"] = torch.nn.functional.cosine_similarity(b.flatten(), r.flatten(), dim=0).item()
        rec["relrmse_64"] = ((b - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
    print(json.dumps(rec), flush=True)
    return rec


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    print("device", torch.cuda.get_device_name(0), flush=True)
    shapes = [  # B Hq Hkv Lq Lk causal dtype
        (1, 2, 2, 256, 256, False, torch.float16), (1, 2, 2, 256, 256, True, torch.float16),
        (1, 4, 2, 300, 300, True, torch.bfloat16), (2, 4, 2, 513, 777, False, torch.bfloat16),
        (1, 2, 1, 1024, 1024, True, torch.float16), (1, 3, 3, 700, 64, False, torch.float16),
        (1, 2, 2, 2048, 2048, True, torch.bfloat16), (1, 1, 1, 129, 1000, True, torch.bfloat16),
    ]
    bad = 0
    for (B, Hq, Hkv, Lq, Lk, causal, dt) in shapes:
        q = torch.randn(B, Hq, Lq, 128, device=DEV).to(dt)
        k = (torch.randn(B, Hkv, Lk, 128, device=DEV) + 1.0).to(dt)
        v = torch.randn(B, Hkv, Lk, 128, device=DEV).to(dt)
        ref = sdpa32(q, k, v, causal)
        tag = f"B{B}H{Hq}/{Hkv}Lq{Lq}Lk{Lk}{'c' if causal else 'nc'}{'f16' if dt == torch.float16 else 'bf16'}"
        r = agree(tag + "/sageattn", lambda: sa.sageattn(q, k, v, is_causal=causal, return_lse=True), ref)
        bad += (not r["finite"]) or r["max_diff_64_vs_128"] > 4e-3 * r["max_o"] + 1e-3
        for gran in ("per_thread", "per_warp"):
            r = agree(tag + f"/fp8_cuda/{gran}", lambda: sa.sageattn_qk_int8_pv_fp8_cuda(
                q, k, v, is_causal=causal, qk_quant_gran=gran, pv_accum_dtype="fp32+fp32", fuse_q_quant=False), ref)
            bad += (not r["finite"]) or r["max_diff_64_vs_128"] > 4e-3 * r["max_o"] + 1e-3
    print("AGREEMENT_FAILURES", bad, flush=True)
    if args.quick:
        return
    # kernel-only and whole-call timings
    sys.path.insert(0, ROOT)
    import bench
    names = ["n1k", "n2k", "n4k", "c3", "n32k", "c3nc"] if args.sweep else ["c3"]
    extra = {"n16k": dict(B=2, H=32, Hkv=32, N=16384, D=128, causal=True, pv="fp8", dtype="bf16", workload="N=16k")}
    bench.CONFIGS.update(extra)
    if args.sweep:
        names.insert(4, "n16k")
    for name in names:
        cfg = bench.CONFIGS[name]
        q, k, v = bench.make_inputs(cfg, DEV, 0)
        ops = bench.prequantize(cfg, q, k, v)
        sm = 1.0 / math.sqrt(cfg["D"])
        fl = bench.flops(cfg)
        rec = dict(config=name)
        for mode in (0, 1, 0, 1):
            lib.sage_set_attn64_mode(mode)
            ms = timeit(lambda: bench.kernel_only_step(cfg, ops, sm), iters=30, warm=10)
            rec.setdefault(f"kernel_ms_mode{mode}", []).append(round(ms, 4))
            rec.setdefault(f"kernel_tflops_mode{mode}", []).append(round(fl / ms / 1e9, 1))
        for mode in (0, 1):
            lib.sage_set_attn64_mode(mode)
            ms = timeit(lambda: bench.e2e_step(cfg, q, k, v), iters=20, warm=5)
            rec[f"e2e_ms_mode{mode}"] = round(ms, 4)
            rec[f"e2e_tflops_mode{mode}"] = round(fl / ms / 1e9, 1)
        print(json.dumps(rec), flush=True)
        del q, k, v, ops
    lib.sage_set_attn64_mode(-1)


if __name__ == "__main__":
    main()
