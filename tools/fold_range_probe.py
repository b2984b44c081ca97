#!/usr/bin/env python3
"""The two FP8 score forms over the magnitude of Q and K: q, k = randn x f (fp16), f = 1 ... 1000 -- i.e. over c, the exponent change per INT8 x INT8 score step
(c = sm_scale log2(e) q_scale k_scale).  The folded form rounds `m + bias c'` once per (row, tile, k scale): an error of up to 0.32 c in the exponent, common to the
group's scores.  This prints, per f: c (median over the groups), whether each form's output is finite, their difference, and each form's distance from fp32 SDPA.

    python tools/fold_range_probe.py [D]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
import sageattention_amd as sa
from sageattention_amd import quant as sq
DEV = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B, H, L = 1, 4, 1000
g = torch.Generator().manual_seed(3)
q0, k0, v = torch.randn(B, H, L, D, generator=g), torch.randn(B, H, L, D, generator=g) + 0.5 * torch.randn(1, H, 1, D, generator=g), torch.randn(B, H, L, D, generator=g).half()
print(f"D {D}, B{B} H{H} L{L}, non-causal and causal; FP8 PV two-level, per-thread scales")
for f in (1, 3, 10, 30, 60, 100, 200, 300, 1000):
    q, k = (q0 * f).half(), (k0 * f).half()
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    km = sq.channel_mean(kd)
    q8, qs, k8, ks = sq.per_thread_int8(qd, kd, km.unsqueeze(2))
    c = float(D ** -0.5 * 1.4426950408889634 * qs.float().median() * ks.float().median())
    for causal in (False, True):
        outs = {}
        for form in ("folded", "exact"):
            o = sa.sageattn_qk_int8_pv_fp8_cuda(qd, kd, vd, is_causal=causal, pv_accum_dtype="fp32+fp32", fp8_scores=form)
            torch.cuda.synchronize()
            outs[form] = o.float().cpu().numpy()
        truth = util.sdpa_f32(q, k, v, causal).numpy()
        sc = float(np.abs(truth).max())
        line = f"f {f:5d}  c {c:9.3e}  {'causal' if causal else 'full  '}"
        for form in ("folded", "exact"):
            o = outs[form]
            fin = np.isfinite(o).all()
            rel = util.rmse(np.nan_to_num(o), truth) / float(np.sqrt((truth ** 2).mean()))
            line += f" | {form}: finite {str(fin):5s} nan {int(np.isnan(o).sum()):7d} rel-RMSE vs SDPA {rel:8.4f}"
        d = np.abs(np.nan_to_num(outs['folded']) - np.nan_to_num(outs['exact'])).max() / sc
        print(line + f" | max|folded - exact| / max|o| {d:9.3e}", flush=True)
