#!/usr/bin/env python3
"""What lies between the kernels of one sageattn() call: run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/call_gaps.py run <cfg>`
(20 whole calls back to back), then `python tools/call_gaps.py parse <dir>` lists, per kernel of the steady-state calls, its duration and the idle time
between its end and the next kernel's start (GPU timestamps of the trace)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import torch
    import bench
    cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c3"]
    dev = torch.device("cuda:0")
    q, k, v = bench.make_inputs(cfg, dev, 5)
    for _ in range(30):
        bench.e2e_step(cfg, q, k, v)
    torch.cuda.synchronize()
    print("done")
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")))
    rows.sort()
    rows = rows[len(rows) // 3:]                      # steady state: the last two thirds
    import collections
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        dur[n0].append((e0 - s0) / 1e3)
        gap[f"{n0} -> {n1}"].append((s1 - e0) / 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    for n, x in dur.items():
        print(f"kernel {n:48s} n={len(x):3d}  median {med(x):8.2f} us")
    for n, x in gap.items():
        print(f"idle   {n:80s} n={len(x):3d}  median {med(x):7.2f} us  (min {min(x):.2f}, max {max(x):.2f})")
    total = (rows[-1][1] - rows[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in rows) / 1e3
    print(f"span {total:.1f} us, kernels {busy:.1f} us, idle {total - busy:.1f} us = {100 * (total - busy) / total:.1f} %")
