#!/usr/bin/env python3
"""FP16-PV causal kernel on bit-exact operands vs the oracle over a list of lengths, 12 launches each: where (which rows) and how reproducibly it differs.
Written for the stress finding of round 5 (seeds 94 / 174 of a 400-seed run; profiles/r5_run_g_stress400_odd_tiles.txt).  SAGE_GFX950_LIB selects a variant library.

    python tools/causal_lengths_probe.py 576 622 739 1100        (DIAG_D=64 for the other head size)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util, oracle
from sageattention_amd import quant as sq, core, _cabi
import test_gpu_parity as T
DEV = torch.device("cuda:0")
D, dt = int(os.environ.get("DIAG_D", "128")), 0
for L in [int(x) for x in sys.argv[1:]]:
    B, Hq, Hkv = 1, 2, 2
    q, k, v = T.rand_qkv(B, Hq, Hkv, L, L, D, dt, seed=174, kbias=0.2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    kmb = util.bits(sq.channel_mean(kd))
    ob, _, aux = oracle.sageattn_dense(util.bits(q), util.bits(k), util.bits(v), dt, is_causal=True, pv="f16", qk_quant_gran="per_thread", km=kmb, warpq=32)
    ref = util.f32(ob, dt)
    q8, qs, gran, q_warp, sm_log2 = core._quant_q(qd, "per_thread", "HND", 32, D ** -0.5)
    _, _, k8, ks, vimg, _, _ = core._prepass_kv(qd, kd, vd, "HND", "per_thread", 64, True, False, False, True, v_fp8=False, v_fp16=True)
    outs = []
    for rep in range(12):
        o, _ = core._attn_dense(False, q8, k8, vimg, None, qs, ks, torch.float16, "HND", True, gran, q_warp, sm_log2, False, False)
        torch.cuda.synchronize()
        outs.append(o.float().cpu().numpy())
    sc = np.abs(ref).max()
    nbad = [int((np.abs(o - ref) > 2e-3 * sc + 2 ** -11 * sc).sum()) for o in outs]
    ndiff = sum(int(not np.array_equal(o, outs[0])) for o in outs[1:])
    rows = sorted(set(int(r) for _, _, r, _ in np.argwhere(np.abs(outs[0] - ref) > 2e-3 * sc + 2 ** -11 * sc)))
    nq = (L + 127) // 128
    print(f"L {L:5d} (D {D}): elements over the bar per launch {nbad[:6]}..., {ndiff} of 11 repeats differ from the first; bad rows {rows[:8]}{'...' if len(rows) > 8 else ''} "
          f"(last block starts at {128 * (nq - 1)})", flush=True)
