#!/usr/bin/env python3
"""Per-workgroup phase times of the one-launch pre-pass (a -DSAGE_PP_TRACE=1 build of sage_prepass.hip).
usage: prepass_trace.py --lib variants/libsage_gfx950_trace.so [--shape B,H,N,D]
Stamps (100 MHz): 0 start, 1 slab loaded, 2 partials published, 3 head complete, 4 head statistics reduced,
5 group maxima (K), 6 stores acknowledged."""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sageattention_amd import _cabi
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--shape", default="2,32,8192,128")
args = ap.parse_args()
_cabi.LIB_PATH = os.path.abspath(args.lib)
lib = _cabi.load()
B, H, L, D = map(int, args.shape.split(","))
dev = torch.device("cuda:0")
k = torch.randn(B, H, L, D, device=dev, dtype=torch.bfloat16)
v = torch.randn(B, H, L, D, device=dev, dtype=torch.bfloat16)
nslab = (L + 511) // 512
nwg = nslab * H * B * 2
used = int(lib.sage_prepass_ws_floats(B, H, L, D))
ws = torch.zeros(used + 16 * nwg + 64, dtype=torch.float32, device=dev)
sync = torch.zeros(int(lib.sage_prepass_sync_words(B, H)), dtype=torch.int32, device=dev)
k8 = torch.empty(B, H, L, D, dtype=torch.int8, device=dev)
ks = torch.empty(B, H, (L + 63) // 64 * 4, dtype=torch.float32, device=dev)
km = torch.empty(B, H, D, dtype=torch.bfloat16, device=dev)
vi = torch.empty(B, H, (L + 63) // 64, D, 64, dtype=torch.uint8, device=dev)
vs = torch.empty(B, H, D, dtype=torch.float32, device=dev)
def run():
    rc = lib.sage_prepass_kv(k.data_ptr(), v.data_ptr(), km.data_ptr(), k8.data_ptr(), ks.data_ptr(), vi.data_ptr(), vs.data_ptr(), None,
                             ws.data_ptr(), sync.data_ptr(), B, H, L, D, *k.stride()[:3], *v.stride()[:3], *k8.stride()[:3],
                             64, _cabi.GRAN_PER_THREAD, _cabi.QSTYLE_TRITON_THREAD, 448.0, 0, _cabi.DTYPE_BF16, None,
                             torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "sage_prepass_kv")
for _ in range(3):
    run()
torch.cuda.synchronize()
run(); torch.cuda.synchronize()
tr = ws[used:used + 16 * nwg].view(torch.int64).cpu().numpy().reshape(nwg, 8)
idx = np.arange(nwg)
part = (idx // (nslab * H)) & 1      # grid order: x slab, y head, z = 2*batch + part
t0 = tr[:, 0].min()
st = (tr[:, :7] - t0) * 0.01          # us
xcc = tr[:, 7] & 0xff
print(f"shape {args.shape}: {nwg} workgroups, kernel span {st[:, 6].max():.1f} us")
for name, sel in (("K", part == 0), ("V", part == 1)):
    s = st[sel]
    d = np.diff(s, axis=1)
    labels = ["load", "stats+publish", "wait for head", "reduce head", "amax pass" if name == "K" else "-", "quantise+store"]
    print(f"-- {name} workgroups ({sel.sum()}): start {s[:,0].mean():7.1f} us (mean)   lifetime {np.mean(s[:,6]-s[:,0]):6.1f} us")
    for j, lab in enumerate(labels):
        col = d[:, j] if not (name == "V" and j == 4) else None
        if col is None:
            continue
        if name == "V" and j == 5:
            col = s[:, 6] - s[:, 4]
        print(f"   {lab:16s} mean {col.mean():7.2f}  p50 {np.median(col):7.2f}  p90 {np.percentile(col, 90):7.2f}  max {col.max():7.2f} us")
# how many workgroups are alive over time (occupancy of the 512 slots)
ts = np.linspace(0, st[:, 6].max(), 9)[1:-1]
alive = [(int(((st[:, 0] <= t) & (st[:, 6] > t)).sum())) for t in ts]
print("alive workgroups at", " ".join(f"{t:.0f}us:{a}" for t, a in zip(ts, alive)))
print("workgroups per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
