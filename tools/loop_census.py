#!/usr/bin/env python3
"""Instruction census of the hottest loop of one kernel in a hipcc -save-temps .s file (tools/asm_stats.sh writes it):
finds the backward branch whose body holds the most MFMAs and counts the body's instructions by kind.
usage: loop_census.py file.s <mangled-kernel-name-substring> [tiles-per-body]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 2           # the pipelined loops are unrolled by two (A / B register sets)
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {l[:-1]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:$", l)}
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i + 1]
        nm = sum("v_mfma" in s for s in seg)
        if best is None or nm > best[0]:
            best = (nm, labels[m.group(1)], i)
nm, a, b = best
seg = [s.split(";")[0].strip() for s in body[a:b + 1]]
seg = [s for s in seg if s and not s.endswith(":") and not s.startswith(".")]
kinds = collections.Counter()
ops = collections.Counter()
for s in seg:
    op = s.split()[0]
    ops[op] += 1
    if op.startswith("v_mfma"): k = "mfma"
    elif op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_cvt_pk_fp8_f32"): k = "valu quarter-rate (exp2, fp8 pack)"
    elif op.startswith("v_"): k = "valu other"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith("global_") or op.startswith("buffer_"): k = "vmem"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): k = "wait / nop / barrier"
    elif op.startswith("s_"): k = "salu / smem / branch"
    else: k = "other"
    kinds[k] += 1
print(f"kernel {key}: loop body lines {a}..{b}, {len(seg)} instructions, {nm} MFMAs = {tiles} tile(s)")
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1]):
    print(f"  {k:38s} {v:5d}   per tile {v / tiles:7.1f}")
print("  VALU by opcode (per tile):", ", ".join(f"{o} {c / tiles:g}" for o, c in sorted(ops.items(), key=lambda kv: -kv[1]) if o.startswith("v_") and not o.startswith("v_mfma")))
