#!/usr/bin/env python3
"""Sections of the key loop per query block, from a trace dump of the FP8 causal kernel (tools/attn_trace.py with SAGE_TRACE_DUMP=file.npy on the
trace build): first scores formed (stamp 8) / six-body trips done (13) / remainder bodies done (14) / last two bodies done (15) / drain + barrier (4)."""
import sys
import numpy as np
t = np.load(sys.argv[1])
t = t[(t[:, 7] != 0) & (t[:, 12] < 4096)]
st = t.astype(np.int64)
base = st[:, 0].min()
us = lambda c: ((st[:, c] - base) & 0xffffffff) / 100.0
qb = t[:, 12].astype(np.int64)
for b in np.unique(qb):
    s = qb == b
    ns = 2 * b                     # steady tiles of a causal block with Lq = Lk
    n6, r = ns // 6, ns % 6
    f = lambda a, c: (us(c)[s] - us(a)[s]).mean()
    print(f"qblk {b:2d}: steady {ns:2d} = {n6} trips + {r} | first tile->scores {f(3, 8):5.2f} | six-body {f(8, 13):6.2f}" + (f" ({f(8, 13) / (6 * n6):.2f}/tile)" if n6 else "") +
          f" | remainder {f(13, 14):5.2f}" + (f" ({f(13, 14) / r:.2f}/tile)" if r else "") + f" | last two bodies {f(14, 15):5.2f} | drain+barrier {f(15, 4):5.2f}"
          f" | entry {us(0)[s].mean():6.2f} exit {us(7)[s].mean():6.2f}")
