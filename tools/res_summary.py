#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel instantiation.
usage: res_summary.py file.res [file2.res ...]"""
import re, sys
for path in sys.argv[1:]:
    rows, cur = [], None
    for l in open(path):
        m = re.search(r"remark:\s+Function Name: (\S+)", l)
        if m:
            cur = {"name": m.group(1)}; rows.append(cur); continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", l)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    print("==", path)
    for r in rows:
        n = r["name"].replace("_ZN4sage16sage_attn_kernel", "attn").replace("EEEvNS_10AttnParamsE", "").replace("ELb", ",b").replace("ELi", ",i").replace("ILi", "<")
        print(f"{n:34s} vgpr {r.get('VGPRs'):>4s} vspill {r.get('VGPRs Spill'):>3s} sspill {r.get('SGPRs Spill'):>3s} scratch {r.get('ScratchSize [bytes/lane]'):>4s} occ {r.get('Occupancy [waves/SIMD]')}")
