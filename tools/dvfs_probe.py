#!/usr/bin/env python3
"""Per-launch duration of the C3 attention kernel over a long back-to-back run (clock / power behaviour).
usage: dvfs_probe.py [n_launches] [config]"""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c3"]
dev = torch.device("cuda:0")
q, k, v = bench.make_inputs(cfg, dev, 1234)
ops = bench.prequantize(cfg, q, k, v)
sm = cfg["D"] ** -0.5
fl = bench.flops(cfg)
for _ in range(3):
    bench.kernel_only_step(cfg, ops, sm)
torch.cuda.synchronize()
time.sleep(2.0)                      # let the device idle / cool
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    bench.kernel_only_step(cfg, ops, sm)
    evs[i + 1].record()
smi = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
torch.cuda.synchronize()
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for i in (0, 1, 2, 3, 5, 10, 20, 50, 100, 200, 300, n - 1):
    if i < n:
        print(f"launch {i:4d}: {d[i]*1e3:7.1f} us  {fl/d[i]/1e9:7.1f} TFLOPS")
print(f"mean first 5: {sum(d[:5])/5*1e3:.1f} us, mean last 50: {sum(d[-50:])/50*1e3:.1f} us")
print("\n".join(l for l in smi.splitlines() if any(t in l for t in ("Power", "sclk", "mclk", "fclk", "Temperature (Sensor junction)", "Temperature (Sensor edge)")))[:1500])
