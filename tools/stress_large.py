#!/usr/bin/env python3
"""One-off stress: tensors with more than 2^31 elements (B=4 H=32 N=131072 D=128, 4.3 GB each in bf16) through
sageattn(); rows at the far end of the address range are checked against fp32 attention computed on slices.
usage: stress_large.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sageattention_amd as sa

B, H, N, D = 5, 32, 131072, 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(B, H, N, D, device=dev, dtype=torch.bfloat16, generator=g)
k = torch.randn(B, H, N, D, device=dev, dtype=torch.bfloat16, generator=g)
v = torch.randn(B, H, N, D, device=dev, dtype=torch.bfloat16, generator=g)
print(f"elements per tensor: {q.numel():,} (> 2^31: {q.numel() > 2**31})")
torch.cuda.synchronize(); t0 = time.perf_counter()
o, lse = sa.sageattn(q, k, v, is_causal=True, return_lse=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sageattn: {dt*1e3:.1f} ms, {4*B*H*N*N*D/2/dt/1e12:.0f} TFLOP/s end to end, finite={bool(torch.isfinite(o).all())}")
worst = 0.0
for (b, h, r0) in [(B - 1, H - 1, N - 128), (B - 1, 0, N // 2), (0, H - 1, 4096), (4, 17, N - 4096), (4, 31, N - 128)]:
    qs = q[b, h, r0:r0 + 128].float()
    s = (qs @ k[b, h, :r0 + 128].float().T) * D ** -0.5
    s = s.masked_fill(torch.arange(r0 + 128, device=dev)[None, :] > (r0 + torch.arange(128, device=dev))[:, None], float("-inf"))
    want = torch.softmax(s, -1) @ v[b, h, :r0 + 128].float()
    got = o[b, h, r0:r0 + 128].float()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0).item()
    rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    dl = (lse[b, h, r0:r0 + 128] - torch.logsumexp(s, -1)).abs().max().item()
    worst = max(worst, 1 - cos)
    print(f"b={b} h={h} rows {r0}..{r0+127}: cos {cos:.6f} rel-rmse {rel:.4f} max|dlse| {dl:.2e}")
assert worst < 2e-3
print("ok")
