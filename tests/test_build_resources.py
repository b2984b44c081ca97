"""Compile-time resource checks of the register-resident kernels (no GPU needed: hipcc cross-compiles for gfx950).

sage_prepass.hip keeps a 512-token slab in registers at 4 waves / SIMD (128 VGPRs).  Its speed depends on the compiler NOT
spilling: every scratch reload sits behind an `s_waitcnt vmcnt(0)` together with the stores in flight, and a version with
~20 spilled registers measured 2.5x slower.  Small source changes tip the allocation over, so the build is checked here."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resource_report(src):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c",
                          os.path.join(ROOT, "sageattention_amd", "csrc", src), "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_prepass_kernel_keeps_its_slab_in_registers():
    kernels = {k: v for k, v in _resource_report("sage_prepass.hip").items() if "prepass_kv_kernel" in k}
    assert len(kernels) == 8, sorted(kernels)                  # D 128 / 64 x fp16 / bf16 x dense / packed (varlen)
    for name, res in kernels.items():
        assert res["VGPRs Spill"] == 0 and res["ScratchSize"] == 0, (name, res)
        assert res["VGPRs"] <= 128 and res["Occupancy"] >= 4, (name, res)        # two 512-thread workgroups per CU


ATTN_UNITS = ("sage_attn_d128_f8.hip", "sage_attn_d128_f8f.hip", "sage_attn_d128_f16.hip", "sage_attn_d64_f8.hip", "sage_attn_d64_f8f.hip",
              "sage_attn_d64_f16.hip")


def _attention_reports():
    """kernel-resource-usage of every instantiation unit of the attention family, compiled side by side."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(ATTN_UNITS)) as ex:
        return dict(zip(ATTN_UNITS, ex.map(_resource_report, ATTN_UNITS)))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_attention_kernels_do_not_spill():
    """The pipelined attention loops pin their instruction order with inline asm and sit a few registers below the occupancy
    limits (D = 128 FP8: 2 waves / SIMD at <= 256 VGPRs; D = 64: 3 waves at <= 168): a spill there would not fail any
    numerical test, only the benchmark.  Every instantiation of every unit -- both FP8 score forms -- is held to zero scratch."""
    reports = _attention_reports()
    kernels = {}
    for unit, rep in reports.items():
        mine = {k: v for k, v in rep.items() if "sage_attn_kernel" in k}
        assert len(mine) == (33 if unit.endswith("f16.hip") else 12), (unit, len(mine))     # (f16: + the packed route's two causal instantiations with the ticket loop, + twelve that read V rows in place: four with the Q quantiser in the prologue, eight on INT8 q)
        kernels.update(mine)
    assert len(kernels) == 2 * (12 + 12 + 33)
    bad = {k: v for k, v in kernels.items() if v["VGPRs Spill"] > 0 or v["ScratchSize"] > 0}
    assert not bad, bad
    d128 = [v for k, v in kernels.items() if "ILi128E" in k]
    assert d128 and all(v["Occupancy"] >= 2 for v in d128)
    d64 = [v for k, v in kernels.items() if "ILi64E" in k]
    assert d64 and all(v["Occupancy"] >= 3 for v in d64), d64
    head = [v for k, v in kernels.items() if "ILi128ELb1ELb1ELb1ELb1E" in k]          # D=128, FP8 PV, causal, per-thread, two-level
    assert len(head) == 6                                                             # INT8 q + fp16 q + bf16 q, folded + exact


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_of_the_library_uses_scratch():
    """Every .hip file that goes into libsage_gfx950.so (the Makefile's SRCS), every kernel: no scratch memory, no spilled VGPR.  A spill is
    invisible to the numerical tests."""
    mk = open(os.path.join(ROOT, "sageattention_amd", "csrc", "Makefile")).read()
    srcs = re.search(r"^SRCS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    assert "sage_attn.hip" in srcs and "sage_prepass.hip" in srcs and all(u in srcs for u in ATTN_UNITS) and len(srcs) >= 14, srcs
    bad = {}
    for src in srcs:
        if src in ATTN_UNITS or src == "sage_prepass.hip":       # (checked above, with their occupancy targets)
            continue
        for name, res in _resource_report(src).items():
            if res.get("ScratchSize", 0) != 0 or res.get("VGPRs Spill", 0) != 0:
                bad[f"{src}:{name}"] = res
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_read_of_an_asm_issued_mfma_result_inside_its_latency():
    """The compiler's hazard recogniser does not see the MFMAs issued from inline asm, and it may move plain copies of their results up to
    just behind the asm statement (round 5: the odd-count rename of the FP16-PV loop, above its nops).  tools/mfma_hazard_lint.py walks the
    listing of every attention unit for such reads."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mfma_hazard_lint as lint
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(lint.UNITS)) as ex:
        results = dict(zip(lint.UNITS, ex.map(lambda u: lint.lint(lint.listing(u)), lint.UNITS)))
    for unit, (findings, n_mfma) in results.items():
        assert n_mfma >= 200, (unit, n_mfma)                   # (the walk did see the pipelined loops)
        assert not findings, (unit, findings[:5])
    # the lint itself: a copy two instructions behind an asm MFMA is reported, one behind enough nops is not
    bad = "_Zk:\n\t;;#ASMSTART\n\tv_mfma_i32_32x32x32_i8 v[96:111], v[0:3], v[4:7], v[96:111]\n\t;;#ASMEND\n\tv_add_f32_e32 v1, v2, v3\n\tv_mov_b32_e32 v64, v97\n"
    ok = bad.replace("v_add_f32_e32 v1, v2, v3", "s_nop 15")
    assert len(lint.lint(bad)[0]) == 1 and len(lint.lint(ok)[0]) == 0
    # ... the other direction (round 6: the loops' MFMAs carry no leading s_nop): a VALU write of an MFMA operand right in front of the MFMA
    bad = "_Zk:\n\tv_mov_b32_e32 v5, v70\n\t;;#ASMSTART\n\tv_mfma_i32_32x32x32_i8 v[96:111], v[0:3], v[4:7], v[96:111]\n\t;;#ASMEND\n"
    ok1 = bad.replace(";;#ASMSTART\n", ";;#ASMSTART\n\ts_nop 1\n")
    ok2 = bad.replace("v_mov_b32_e32 v5, v70", "v_mov_b32_e32 v8, v70")
    assert len(lint.lint(bad)[0]) == 1 and len(lint.lint(ok1)[0]) == 0 and len(lint.lint(ok2)[0]) == 0
    # ... and a VALU read of a transcendental's result in the next issue slot
    bad = "_Zk:\n\t;;#ASMSTART\n\tv_exp_f32 v3, v3\n\tv_add_f32 v1, v1, v3\n\t;;#ASMEND\n"
    ok = "_Zk:\n\t;;#ASMSTART\n\tv_exp_f32 v3, v3\n\tv_exp_f32 v4, v4\n\tv_add_f32 v1, v1, v3\n\t;;#ASMEND\n"
    assert len(lint.lint(bad)[0]) == 1 and len(lint.lint(ok)[0]) == 0
