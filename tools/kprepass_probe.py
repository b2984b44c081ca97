import sys, torch
sys.path.insert(0, '.')
from sageattention_amd import quant as sq, _cabi
dev = torch.device("cuda:0")
def med(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); t.append(a.elapsed_time(b) * 1e3)
    return sorted(t)[len(t) // 2]
for (B, H, N, D) in ((2, 32, 4096, 128), (2, 32, 2048, 128), (2, 32, 1024, 128), (2, 32, 8192, 128), (2, 32, 16384, 128), (2, 48, 4096, 64)):
    k = torch.randn(B, H, N, D, device=dev).half()
    def seq():
        km = sq.channel_mean(k)
        return sq._quant(k, km, 64, 64, _cabi.GRAN_PER_THREAD, True, _cabi.QSTYLE_TRITON_THREAD, 1.0, "HND", 4)
    one = lambda: sq.prepass_kv_fp8(k, None, "HND", smooth_k=True, qk_quant_gran="per_thread", v_fp16=True)
    a = one(); b = seq()
    assert torch.equal(a[1], b[0]) and torch.equal(a[2], b[1])
    print(f"B{B} H{H} N{N} D{D}: K-only one launch {med(one):7.1f} us   mean + quantiser sequence {med(seq):7.1f} us", flush=True)
