#!/usr/bin/env python3
"""Per query block (word 12 of a trace record) phase times of one traced launch: tools/attn_trace.py with SAGE_TRACE_DUMP=file.npy, then
trace_clusters.py file.npy  -- which workgroups of a causal launch are the slow ones, and who they share their CU with."""
import sys
import numpy as np
t = np.load(sys.argv[1])
st = t[:, :8].astype(np.int64); st = (st - st[:, :1].min()) & 0xffffffff; us = st / 100.0
kl = us[:, 4] - us[:, 3]; ent = us[:, 0]; ex = us[:, 7]; qb = t[:, 12].astype(np.int64)
hw, xcc = t[:, 9].astype(np.int64), t[:, 10].astype(np.int64) & 0xf
cu = (xcc << 16) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 0xf)
for b in np.unique(qb):
    s = qb == b
    # the co-resident workgroups: same CU, lifetimes overlapping
    mates = []
    for i in np.nonzero(s)[0][:64]:
        m = (cu == cu[i]) & (ent < ex[i]) & (ex > ent[i]); m[i] = False
        mates.extend(qb[m].tolist())
    mates = np.bincount(np.array(mates, dtype=np.int64), minlength=qb.max() + 1) if mates else []
    print(f"qblk {b:3d} ({2 * b + 2:3d} tiles): key loop mean {kl[s].mean():7.2f} min {kl[s].min():7.2f} max {kl[s].max():7.2f} | entry {ent[s].mean():6.2f} exit {ex[s].mean():7.2f}"
          f" | pre {(us[s, 3] - us[s, 0]).mean():5.2f} post {(us[s, 7] - us[s, 4]).mean():5.2f} | CU mates by qblk {list(mates)}")
