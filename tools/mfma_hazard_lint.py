#!/usr/bin/env python3
"""Lint of the attention units' ISA for reads of inline-asm MFMA results inside the MFMA's latency.

The pipelined loops of sage_attn_kernel.h issue their MFMAs from inline asm, so the compiler's hazard recogniser does not see them: it adds
no wait states in front of an instruction that reads (or overwrites) their destination registers, and it is free to move plain register
copies of those results up to just behind the asm statement.  Round 5 found such a copy (the odd-count rename sA = sB of the FP16-PV loop,
hoisted above its nops in the causal D = 128 instantiations only; wrong rows in the last query block of lengths with an odd number of
pipelined tiles).  This script walks the compiler's listing of a unit and reports every instruction that touches the destination of an
asm-issued MFMA fewer wait states behind it than the ISA's XDL -> VALU rule asks (8-pass MFMA: 11, 16-pass: 19), except another MFMA
accumulating into the very same registers.  Straight-line approximation: it follows the listing, not the branches.

    python tools/mfma_hazard_lint.py [unit.hip ...]        (default: the six attention units)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sageattention_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
UNITS = ("sage_attn_d128_f8.hip", "sage_attn_d128_f8x.hip", "sage_attn_d128_f16.hip", "sage_attn_d64_f8.hip", "sage_attn_d64_f8x.hip",
         "sage_attn_d64_f16.hip")
NEED = {"v_mfma_f32_32x32x64_f8f6f4": 19, "v_mfma_scale_f32_32x32x64_f8f6f4": 19}        # 16 passes; everything else used here: 8 passes
NEED_DEFAULT = 11
_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def listing(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "unit.s")
        r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return open(out).read()


def _regs(text):
    return [(int(a), int(b)) if a else (int(c), int(c)) for a, b, c in _REG.findall(text)]


def lint(asm_text):
    """-> (list of findings, number of asm-issued MFMAs seen).  A finding: (kernel, line number, instruction, the MFMA, wait states short)."""
    findings, n_mfma = [], 0
    kernel, in_asm, pending = "?", False, []           # pending: [lo, hi, wait states still needed, text]
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending = m.group(1), []
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line[0] in ".;" or line.endswith(":") or not raw.startswith("\t"):
            continue
        line = line.split(";")[0].strip()
        if not line:
            continue
        op, _, rest = line.partition(" ")
        if op.startswith("s_") and not _REG.search(rest):
            states = int(rest.strip() or 0) + 1 if op == "s_nop" else 1
            if op in ("s_endpgm", "s_setpc_b64"):
                pending = []
        else:
            ops = [o.strip() for o in rest.split(",")]
            is_mfma = op.startswith("v_mfma")
            for lo, hi, need, what in pending:
                if need <= 0:
                    continue
                for i, o in enumerate(ops):
                    for a, b in _regs(o):
                        if a <= hi and b >= lo:
                            if is_mfma and i in (0, 3) and (a, b) == (lo, hi):      # accumulates into the same registers: hardware interlock
                                continue
                            findings.append((kernel, ln, line, what, need))
            states = 1
            if is_mfma and in_asm:
                n_mfma += 1
                d = _regs(ops[0])
                if d:
                    pending.append([d[0][0], d[0][1], NEED.get(op, NEED_DEFAULT) + states, line])
        for p in pending:
            p[2] -= states
        pending = [p for p in pending if p[2] > 0]
    return findings, n_mfma


def main(argv):
    units = argv or UNITS
    total = 0
    for u in units:
        f, n = lint(listing(u))
        print(f"{u}: {n} asm-issued MFMAs, {len(f)} reads inside their latency")
        for kernel, ln, ins, what, need in f[:40]:
            print(f"  {kernel[:90]} line {ln}: `{ins}` is {need} wait state(s) short of `{what}`")
        total += len(f)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
