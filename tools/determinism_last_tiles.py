import os, sys, torch
sys.path.insert(0, os.getcwd())
import sageattention_amd as sa
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
def mk(B, H, Lq, Lk, D, dt):
    return (torch.randn(B, H, Lq, D, device=dev, generator=g).to(dt), torch.randn(B, H, Lk, D, device=dev, generator=g).to(dt), torch.randn(B, H, Lk, D, device=dev, generator=g).to(dt))
cases = [
    ("fp8 D128 N1024 causal", sa.sageattn, mk(2, 32, 1024, 1024, 128, torch.bfloat16), dict(is_causal=True)),
    ("fp8 D128 N2048 causal", sa.sageattn, mk(2, 32, 2048, 2048, 128, torch.bfloat16), dict(is_causal=True)),
    ("fp8 D64 N2048 causal", sa.sageattn, mk(2, 32, 2048, 2048, 64, torch.float16), dict(is_causal=True)),
    ("fp8 D128 Lk 1000 non-causal (ragged tail)", sa.sageattn, mk(2, 32, 1000, 1000, 128, torch.bfloat16), dict(is_causal=False)),
    ("fp8 D64 N17776 non-causal (C5, tickets, ragged tail)", sa.sageattn, mk(1, 16, 17776, 17776, 64, torch.bfloat16), dict(is_causal=False)),
    ("fp8 D128 Lq 4096 Lk 512 non-causal (cross)", sa.sageattn, mk(2, 32, 4096, 512, 128, torch.bfloat16), dict(is_causal=False)),
    ("fp16-PV D128 N2048 causal", sa.sageattn_qk_int8_pv_fp16_cuda, mk(2, 32, 2048, 2048, 128, torch.float16), dict(is_causal=True)),
    ("fp16-PV D128 N2048 non-causal", sa.sageattn_qk_int8_pv_fp16_cuda, mk(2, 32, 2048, 2048, 128, torch.bfloat16), dict(is_causal=False)),
    ("triton api D128 N2048 causal", sa.sageattn_qk_int8_pv_fp16_triton, mk(2, 32, 2048, 2048, 128, torch.float16), dict(is_causal=True)),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
s2 = torch.cuda.Stream()
for name, fn, (q, k, v), kw in cases:
    ref = fn(q, k, v, **kw)
    bad = 0
    for i in range(reps):
        if i & 1:
            with torch.cuda.stream(s2):
                o = fn(q, k, v, **kw)
            s2.synchronize()
        else:
            o = fn(q, k, v, **kw)
        if not torch.equal(o, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name:55s} differing calls {bad:3d} / {reps}   finite {bool(torch.isfinite(ref.float()).all())}")
