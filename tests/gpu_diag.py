#!/usr/bin/env python3
"""GPU diagnostic dump (not a test): runs the smallest cases through every kernel and saves HIP and oracle
results side by side into gpurun_out/diag_*.npz so that a layout bug can be analysed off-box."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
import oracle
import sageattention_amd as sa
from sageattention_amd import quant as sq

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
dev = torch.device("cuda:0")
props = torch.cuda.get_device_properties(0)
print("device:", props.name, getattr(props, "gcnArchName", "?"), "CUs", props.multi_processor_count, "cores", os.cpu_count())


def run(tag, B, Hq, Hkv, Lq, Lk, D, dt, causal):
    tt = torch.float16 if dt == 0 else torch.bfloat16
    g = torch.Generator().manual_seed(42)
    q = torch.randn(B, Hq, Lq, D, generator=g).to(tt)
    k = (torch.randn(B, Hkv, Lk, D, generator=g) + 1.0).to(tt)
    v = torch.randn(B, Hkv, Lk, D, generator=g).to(tt)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    km = sq.channel_mean(kd).unsqueeze(2)
    save = dict(q=util.bits(q), k=util.bits(k), v=util.bits(v), km=util.bits(km))
    for gran in ("per_block", "per_warp", "per_thread"):
        for pv in ("f8", "f16"):
            try:
                fn = sa.sageattn_qk_int8_pv_fp8_cuda if pv == "f8" else sa.sageattn_qk_int8_pv_fp16_cuda
                acc = "fp32+fp32" if pv == "f8" else "fp32"
                o, lse = fn(qd, kd, vd, is_causal=causal, qk_quant_gran=gran, pv_accum_dtype=acc, return_lse=True)
                torch.cuda.synchronize()
                ro, rlse, aux = oracle.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv=pv,
                                                      qk_quant_gran=gran, return_lse=True, km=util.bits(km[:, :, 0]))
                got, ref = o.float().cpu().numpy(), util.f32(ro, dt)
                err = np.abs(got - ref)
                print(f"{tag:28s} {gran:10s} {pv:3s} max|d|={err.max():.3e} max|o|={np.abs(ref).max():.3e} "
                      f"nan={np.isnan(got).sum()} lse_d={np.abs(lse.cpu().numpy() - rlse).max():.3e}")
                save[f"o_{gran}_{pv}"] = got
                save[f"ref_{gran}_{pv}"] = ref
            except Exception:
                print(f"{tag} {gran} {pv}: EXCEPTION")
                traceback.print_exc()
    # raw kernel products for the per_warp path
    try:
        q8, qs, k8, ks = sq.per_warp_int8(qd, kd, km)
        img8, vs, _ = sq.per_channel_fp8(vd)
        img16 = sq.prep_v_fp16(vd)
        torch.cuda.synchronize()
        save.update(q8=q8.cpu().numpy(), qs=qs.cpu().numpy(), k8=k8.cpu().numpy(), ks=ks.cpu().numpy(),
                    img8=img8.cpu().numpy(), vs=vs.cpu().numpy(), img16=img16.cpu().view(torch.int16).numpy())
    except Exception:
        traceback.print_exc()
    np.savez_compressed(os.path.join(OUT, f"diag_{tag}.npz"), **save)


run("tile_d128_f16", 1, 1, 1, 128, 64, 128, 0, False)
run("tile_d64_f16", 1, 1, 1, 128, 64, 64, 0, False)
run("two_tiles_d128_causal", 1, 1, 1, 128, 128, 128, 0, True)
run("ragged_gqa_d128_bf16_causal", 1, 4, 2, 300, 300, 128, 1, True)
print("diag done")
