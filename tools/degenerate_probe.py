#!/usr/bin/env python3
"""tests/test_gpu_parity.py::test_degenerate_inputs_vs_oracle as a table: per (api, input kind, causal) the output / LSE differences against the oracle."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util, oracle
import sageattention_amd as sa
from sageattention_amd import quant as sq
import test_gpu_parity as T
DEV = torch.device("cuda:0")
kinds = sys.argv[1:] or T.DEGENERATE
for what in kinds:
    for api in ("f8", "f8x", "f16", "triton"):
        D = 128 if api != "triton" else 64
        if what.startswith("scale"):
            q, k, v, dt = T.degenerate_qkv("q_zero", D)
            q, k, _ = T.rand_qkv(1, 4, 2, 330, 330, D, dt, seed=61, kbias=1.0)
            q, k = (q.float() * float(what[5:])).half(), (k.float() * float(what[5:])).half()
        else:
            q, k, v, dt = T.degenerate_qkv(what, D)
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        km = util.bits(sq.channel_mean(kd))
        for causal in (False, True):
            if api == "triton":
                ref, lse_ref, aux = oracle.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f16_triton", qk_quant_gran="per_block", return_lse=True, km=km)
                o, lse = sa.sageattn_qk_int8_pv_fp16_triton(qd, kd, vd, is_causal=causal, return_lse=True)
            else:
                form = "exact" if api == "f8x" else "folded"
                ref, lse_ref, aux = oracle.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=causal, pv="f8" if api.startswith("f8") else "f16",
                                                          qk_quant_gran="per_thread", return_lse=True, km=km, fp8_scores=form)
                if api.startswith("f8"):
                    o, lse = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_thread", pv_accum_dtype="fp32+fp32", return_lse=True, fp8_scores=form)
                else:
                    o, lse = sa.sageattn_qk_int8_pv_fp16_cuda(qd, kd, vd, is_causal=causal, qk_quant_gran="per_thread", pv_accum_dtype="fp32", return_lse=True)
            torch.cuda.synchronize()
            got, want = o.float().cpu().numpy(), util.f32(ref, dt)
            lg = lse.cpu().numpy()
            truth = util.sdpa_f32(q, k, v, causal).numpy()
            sc = float(np.nanmax(np.abs(want))) if np.isfinite(want).any() else float("nan")
            c = float(D ** -0.5 * 1.4427 * np.median(aux["qs"]) * np.median(aux["ks"])) if api != "triton" else float(np.median(aux["qs"]) * np.median(aux["ks"]))
            print(f"{what:12s} {api:6s} {'causal' if causal else 'full  '} c {c:9.2e} max|o| {sc:9.3e} | kernel nan {int(np.isnan(got).sum()):6d} oracle nan {int(np.isnan(want).sum()):6d} | "
                  f"max|kernel - oracle| {np.nanmax(np.abs(np.nan_to_num(got) - np.nan_to_num(want))):9.3e} | lse diff {np.nanmax(np.abs(np.nan_to_num(lg, posinf=1e30, neginf=-1e30) - np.nan_to_num(lse_ref, posinf=1e30, neginf=-1e30))):9.3e} "
                  f"(|lse| max {np.nanmax(np.abs(np.nan_to_num(lse_ref, posinf=1e30))):8.2e}) | kernel vs SDPA {np.abs(np.nan_to_num(got) - truth).max():9.3e}, oracle vs SDPA {np.abs(np.nan_to_num(want) - truth).max():9.3e}", flush=True)
