#!/usr/bin/env python3
"""Lint of the attention units' ISA for reads of inline-asm MFMA results inside the MFMA's latency.

The pipelined loops of sage_attn_kernel.h issue their MFMAs from inline asm, so the compiler's hazard recogniser does not see them: it adds
no wait states in front of an instruction that reads (or overwrites) their destination registers, and it is free to move plain register
copies of those results up to just behind the asm statement.  Round 5 found such a copy (the odd-count rename sA = sB of the FP16-PV loop,
hoisted above its nops in the causal D = 128 instantiations only; wrong rows in the last query block of lengths with an odd number of
pipelined tiles).  This script walks the compiler's listing of a unit and reports every instruction that touches the destination of an
asm-issued MFMA fewer wait states behind it than the ISA's XDL -> VALU rule asks (8-pass MFMA: 11, 16-pass: 19), except another MFMA
accumulating into the very same registers; and every non-transcendental instruction that reads the result of an asm-issued
transcendental (v_exp_f32 ...) in the very next issue slot.  Back-to-back MFMAs are modelled as the matrix pipe issues them (one per `passes` issue
slots: the wait in front of the second one counts for the first).  It follows the listing and, where MFMAs are still pending at a branch, the branch target too.

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
UNITS = ("sage_attn_d128_f8.hip", "sage_attn_d128_f8f.hip", "sage_attn_d128_f16.hip", "sage_attn_d64_f8.hip", "sage_attn_d64_f8f.hip",
         "sage_attn_d64_f16.hip")
NEED = {"v_mfma_f32_32x32x64_f8f6f4": 19, "v_mfma_scale_f32_32x32x64_f8f6f4": 19}        # 16 passes; everything else used here: 8 passes
NEED_DEFAULT = 11
PASSES = {k: 16 for k in NEED}
PASSES_DEFAULT = 8
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")      # their result needs one issue slot before a non-transcendental VALU reads it
TAKEN_BRANCH = 2           # issue slots a taken branch costs on top of its own (assumed: the instruction buffer refills; >= 8 clocks)
_REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def listing(src):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "unit.s")
        r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", *os.environ.get("SAGE_LINT_FLAGS", "").split(), "-S",
                            os.path.join(CSRC, src), "-o", out],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return open(out).read()


def _regs(text):
    return [(int(a), int(b)) if a else (int(c), int(c)) for a, b, c in _REG.findall(text)]


def _parse(asm_text):
    """-> list of (kernel, line number, text, op, operands, issued from inline asm), {label: index of the instruction behind it}"""
    ins, labels = [], {}
    kernel, in_asm = "?", False
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            labels[m.group(1)] = len(ins)
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
        ins.append((kernel, ln, line, op, [o.strip() for o in rest.split(",")] if rest.strip() else [], in_asm))
    return ins, labels


def _walk(ins, labels, start, pending, findings, top):
    """Follows the listing from instruction `start`.  top: the main pass (records new asm MFMAs, forks at branches into their targets);
    a fork only carries the MFMAs pending at its branch and ends when their windows have passed.  -> number of asm MFMAs seen"""
    n_mfma, kernel = 0, ins[start][0] if start < len(ins) else "?"
    busy = 0                   # passes the matrix pipe is still occupied for: an MFMA issued inside them waits for the remainder
    trans = None               # (register, text) of an asm-issued transcendental in the previous issue slot
    for idx in range(start, len(ins)):
        k, ln, line, op, ops, in_asm = ins[idx]
        if k != kernel:
            if not top:
                return n_mfma
            kernel, pending = k, []
        if op.startswith("s_") and not any(_REG.search(o) for o in ops):
            states = int(ops[0] or 0) + 1 if op == "s_nop" and ops else 1
            if op in ("s_endpgm", "s_setpc_b64"):
                pending = []
            if top and pending and (op.startswith("s_cbranch") or op == "s_branch") and ops and ops[0] in labels:
                _walk(ins, labels, labels[ops[0]], [[p[0], p[1], p[2] - states - TAKEN_BRANCH, p[3]] for p in pending], findings, False)
                if op == "s_branch":
                    pending = []
            trans = None
        else:
            is_mfma = op.startswith("v_mfma")
            if trans is not None and not op.startswith(TRANS) and top:
                for o in ops[1:] if op.startswith("v_") else ops:
                    if any(a <= trans[0] <= b for a, b in _regs(o)):
                        findings.add((kernel, ln, line, trans[1], 1))
            trans = None
            if in_asm and op.startswith(TRANS) and ops and _regs(ops[0]):
                trans = (_regs(ops[0])[0][0], line)
            for lo, hi, need, what in pending:
                if need <= 0:
                    continue
                for i, o in enumerate(ops):
                    for a, b in _regs(o):
                        if a <= hi and b >= lo:
                            if is_mfma and i in (0, 3) and (a, b) == (lo, hi):      # accumulates into the same registers: hardware interlock
                                continue
                            findings.add((kernel, ln, line, what, need))
            states = 1
            if is_mfma:
                for p in pending:               # (the stall in front of this MFMA counts for the MFMAs already issued, not for this one)
                    p[2] -= busy
                busy = PASSES.get(op, PASSES_DEFAULT) + 1
                if in_asm and top:
                    n_mfma += 1
                    d = _regs(ops[0])
                    if d:
                        pending.append([d[0][0], d[0][1], NEED.get(op, NEED_DEFAULT) + states, line])
        busy = max(0, busy - states)
        for p in pending:
            p[2] -= states
        pending = [p for p in pending if p[2] > 0]
        if not top and not pending:
            return n_mfma
    return n_mfma


VALU_TO_MFMA = 2           # wait states between a VALU write of a VGPR and an MFMA reading it as SrcA / SrcB / SrcC (cdna_hip_programming.md 5.7 item 2)


def _valu_to_mfma(ins, findings):
    """The other direction: an asm-issued MFMA whose operand a VALU instruction wrote fewer than VALU_TO_MFMA wait states before it.  The
    pipelined loops issue their MFMAs without a leading s_nop (round 6: 12-24 nops per tile); that is sound only while nothing -- an asm
    statement of ours or a copy the compiler places in front of one -- writes an operand right before the MFMA.  Follows the listing
    (fall-through order; a label keeps the history: the body of a loop starts far from its first MFMA)."""
    recent = []                # [wait states since, lo, hi, text] of VALU writes
    kernel = None
    for k, ln, line, op, ops, in_asm in ins:
        if k != kernel:
            kernel, recent = k, []
        states = int(ops[0] or 0) + 1 if op == "s_nop" and ops else 1
        if op.startswith("v_mfma"):
            if in_asm:
                for o in ops[1:4]:
                    for a, b in _regs(o):
                        for since, lo, hi, what in recent:
                            if a <= hi and b >= lo and since < VALU_TO_MFMA:
                                findings.add((kernel, ln, line, what + "  [VALU write -> MFMA operand]", VALU_TO_MFMA - since))
        elif op.startswith("v_") and ops and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            d = _regs(ops[0])
            if d:
                recent.append([-1, d[0][0], d[0][1], line])          # (-1: the increment below makes it 0 for the next instruction)
            if op.startswith("v_permlane") and len(ops) > 1 and _regs(ops[1]):
                recent.append([-1, _regs(ops[1])[0][0], _regs(ops[1])[0][1], line])
        for r in recent:
            r[0] += states
        recent = [r for r in recent if r[0] < VALU_TO_MFMA]


def lint(asm_text):
    """-> (list of findings, number of asm-issued MFMAs seen).  A finding: (kernel, line number, instruction, the MFMA, wait states short).
    The walk follows the listing and, with MFMAs pending at a branch, also the branch's target (loop back-edges, skipped blocks)."""
    ins, labels = _parse(asm_text)
    findings = set()
    n = _walk(ins, labels, 0, [], findings, True) if ins else 0
    _valu_to_mfma(ins, findings)
    return sorted(findings, key=lambda f: f[1]), n


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
