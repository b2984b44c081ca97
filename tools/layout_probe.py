"""NHD vs HND whole-call timing of sageattn() at the C3 shape (K-row stride sensitivity probe)."""
import sys, time, torch
sys.path.insert(0, ".")
import sageattention_amd as sa

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

B, H, N, D = 2, 32, 8192, 128
for layout in ("HND", "NHD"):
    shp = (B, H, N, D) if layout == "HND" else (B, N, H, D)
    q, k, v = (torch.randn(shp, device="cuda", dtype=torch.float16) for _ in range(3))
    for causal in (False, True):
        dt = t(lambda: sa.sageattn(q, k, v, tensor_layout=layout, is_causal=causal))
        fl = 4.0 * B * H * N * N * D / (2 if causal else 1)
        print(f"{layout} causal={causal}: {dt*1e3:.3f} ms  {fl/dt/1e12:.0f} TFLOPS")
