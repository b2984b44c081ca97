#!/usr/bin/env python3
"""Cold-clock determinism of the FP16-PV route's components: sleep between launches so every call starts on an idle device."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa
from sageattention_amd import quant as sq, core
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(2)
B, H, N, D = 2, 32, 4096, 128
q, k, v = [torch.randn(B, H, N, D, device=dev, dtype=torch.float32, generator=g).to(torch.float16) for _ in range(3)]
ref_pp = sq.prepass_kv_fp8(k, None, "HND", smooth_k=False)
vimg = sq.prep_v_fp16(v)
k8, ks = ref_pp[1], ref_pp[2]
sm = core._sm_log2(D ** -0.5)
ref_o, _ = core._attn_fused_q(q, k8, vimg, None, ks, "HND", True, sm, False)
torch.cuda.synchronize()
bad_pp = bad_at = bad_e2e = 0
ref_e = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, smooth_k=False)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
    pp = sq.prepass_kv_fp8(k, None, "HND", smooth_k=False)
    torch.cuda.synchronize()
    if not (torch.equal(pp[1], k8) and torch.equal(pp[2], ks)):
        bad_pp += 1
    time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
    o, _ = core._attn_fused_q(q, k8, vimg, None, ks, "HND", True, sm, False)
    torch.cuda.synchronize()
    if not torch.equal(o, ref_o):
        bad_at += 1
        d = (o != ref_o).nonzero()
        print("attention differs:", int(len(d)), "elements; b", d[:, 0].unique().tolist(), "h", d[:, 1].unique().tolist()[:8], "rows", int(d[:, 2].min()), int(d[:, 2].max()),
              "cols", int(d[:, 3].min()), int(d[:, 3].max()), "max|diff|", float((o.float() - ref_o.float()).abs().max()))
    time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
    e = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, smooth_k=False)
    torch.cuda.synchronize()
    if not torch.equal(e, ref_e):
        bad_e2e += 1
print(f"cold-start calls differing from the warm reference: pre-pass {bad_pp}, attention {bad_at}, whole call {bad_e2e}")
