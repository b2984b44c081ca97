#!/usr/bin/env python3
"""V -> 2V must double the FP16-PV output bit-exactly: repeat with fresh allocations, report where it does not."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, H, N, D = 2, 32, 4096, 128
q, k, v = [torch.randn(B, H, N, D, device=dev, dtype=torch.float32, generator=g).to(torch.float16) for _ in range(3)]
v = torch.where(v.abs() < 2.0 ** -10, torch.zeros_like(v), v)
for extra in (dict(), dict(fuse_q_quant=False), dict(fused_prepass=False), dict(fuse_q_quant=False, fused_prepass=False)):
    bad = 0
    info = ""
    junk = []
    for rep in range(25):
        junk.append(torch.full((1 + rep * 37, 1024), float("nan"), device=dev))        # perturb the allocator, poison freed blocks
        if len(junk) > 3:
            junk.pop(0)
        o = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v, is_causal=True, smooth_k=False, **extra)
        o2 = sa.sageattn_qk_int8_pv_fp16_cuda(q, k, v * 2, is_causal=True, smooth_k=False, **extra)
        normal = o.abs() >= 2.0 ** -13
        ne = (o2 != o * 2) & normal
        if bool(ne.any()):
            bad += 1
            idx = ne.nonzero()
            b0, h0, r0, d0 = idx[0].tolist()
            info = (f"first at b{b0} h{h0} row{r0} d{d0}: {o2[b0,h0,r0,d0].item()} vs {2*o[b0,h0,r0,d0].item()}; {int(ne.sum())} elements, "
                    f"rows {int(idx[:,2].min())}..{int(idx[:,2].max())}, heads {sorted(set((idx[:,0]*H+idx[:,1]).tolist()))[:6]}, nan {bool(o2.isnan().any())}")
    print(f"{str(extra):55s} failing reps {bad:2d} / 25  {info}")
