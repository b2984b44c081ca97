#!/usr/bin/env python3
"""PMC evidence at HEAD (rounds 5-6; ROUND below names the output files): one set of rocprofv3 --pmc passes per dominant kernel instantiation, summarised with ONE formula per
derived figure into gpurun_out/<tag>/r6_pmc_<cfg>.txt and gpurun_out/<tag>/r6_pmc.json (copied to profiles/ and read by bench.py).

Passes per configuration (a process each; counters only, no trace domain beside --pmc):
  A  GRBM_GUI_ACTIVE + 8 SQ slots   time base, resident waves, VALU-active, MFMA-busy, waits
  B  GRBM_GUI_ACTIVE + 8 SQ slots   instruction counts, LDS bank conflicts
  F  FETCH_SIZE   (alone)           what the L2s request from the fabric, KiB
  W  WRITE_SIZE   (alone)           KiB
HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: MI355X_MICROARCH.md, HBM section -- on gfx950 FETCH_SIZE reports half the
bytes of 16-B-per-lane streaming reads (what the attention kernel's LDS-DMA tile loads are); the pre-pass reads 8 B per lane, for which the
factor was calibrated at 2.0 as well (profiles/r2_run_r3j_pmc_prepass_c3.txt: a load-only build reading a known 268.4 MB).
usage (on the GPU box): pmc_collect.py <tag> [cfg ...]      cfg: c2 c3 c4 c4nc c5 c2t pp"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r6"
GROUPS = {
    "A": "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES",
    "B": "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS",
    "F": "FETCH_SIZE",
    "W": "WRITE_SIZE",
}
TARGETS = {   # cfg -> (command, kernel-name filter, what it is, algorithmic bytes per launch)
    "c3": (["tools/run_kernel.py", "c3", "3"], "sage_attn_kernel", "C3 attention kernel (B2 H32 N8192 D128 causal, INT8-q per-thread, FP8 PV two-level, exact scores: the default)", 335.5e6),
    "c2": (["tools/run_kernel.py", "c2", "3"], "sage_attn_kernel", "C2 attention kernel (B2 H32 N4096 D128 causal, INT8-q per-thread, FP16 PV: exact scores, lazily refreshed reference)", 201.3e6),
    "c2t": (["tools/run_kernel.py", "c2t", "3"], "sage_attn_kernel", "Triton-named API at the C2 shape: attention kernel with the per-block Q quantiser in its prologue (fp16 q read: 2 B/elt)", 268.4e6),
    "c4": (["tools/run_kernel.py", "c4", "3"], "sage_attn_kernel", "C4 causal packed attention launch over the work list (persistent since round 5)", 651.9e6),
    "c4nc": (["tools/run_kernel.py", "c4nc", "3"], "sage_attn_kernel", "C4 non-causal packed attention launch over the work list (persistent)", 651.9e6),
    "c5": (["tools/run_kernel.py", "c5", "3"], "sage_attn_kernel", "C5 attention kernel (B2 H48 N17776 D64 non-causal, persistent launch, exact scores: the default)", 546.1e6),
    "c3f": (["tools/run_kernel.py", "c3", "3", "folded"], "sage_attn_kernel", "C3 attention kernel, the opt-in FOLDED score variant (fp8_scores=\"folded\")", 335.5e6),
    "c5f": (["tools/run_kernel.py", "c5", "3", "folded"], "sage_attn_kernel", "C5 attention kernel, the opt-in FOLDED score variant", 546.1e6),
    "c2r": (["tools/run_kernel.py", "c2r", "3"], "sage_attn_kernel", "C2 default route on fp16 inputs: fused per-thread Q quantisation + V rows in place (fp16 q and v read: 2 B/elt each)", 234.9e6),
    "pp": (["tools/run_prepass.py", "fused", "2,32,8192,128", "3"], "prepass_kv_kernel", "one-launch K / V pre-pass at the C3 shape (Infinity Cache flushed between launches)", 402.7e6),
}


def one_pass(out, cfg, grp):
    cmd, filt, _, _ = TARGETS[cfg]
    d = os.path.join(out, f"tmp_{cfg}_{grp}")
    shutil.rmtree(d, ignore_errors=True)
    log = os.path.join(out, f"pass_{cfg}_{grp}.log")
    with open(log, "w") as f:
        subprocess.run(["timeout", "150", "rocprofv3", "--pmc", *GROUPS[grp].split(), "--output-format", "csv", "-d", d, "--", sys.executable, *cmd],
                       stdout=f, stderr=subprocess.STDOUT, cwd=ROOT)
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if filt in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    if acc:
        os.remove(log)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def summarise(cfg, raw):
    g = lambda k: raw[k][0] if k in raw else float("nan")
    gui = g("GRBM_GUI_ACTIVE")                      # summed over the 8 XCDs
    simd_cycles = 1024.0 * gui / 8.0                # 256 CUs x 4 SIMDs, elapsed cycles each
    d = collections.OrderedDict()
    d["elapsed_cycles"] = gui / 8.0
    d["waves_per_simd"] = 4.0 * g("SQ_WAVE_CYCLES") / simd_cycles            # SQ_WAVE_CYCLES counts quad-cycles
    d["valu_active"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles
    d["mfma_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles
    d["wait_inst_any_of_wave_time"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    d["wait_any_of_wave_time"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    d["mfma_insts_per_wave"] = g("SQ_INSTS_MFMA") / g("SQ_WAVES")
    d["lds_bank_conflict_of_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    d["fetch_kib"] = g("FETCH_SIZE")
    d["write_kib"] = g("WRITE_SIZE")
    d["traffic_bytes"] = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
    d["algorithmic_bytes"] = TARGETS[cfg][3]
    d["traffic_over_algorithmic"] = d["traffic_bytes"] / TARGETS[cfg][3]
    return d


HEADER = """# rocprofv3 --pmc passes over: {what}
# tools/pmc_collect.py at commit {head} (mean per launch of the named kernel, n = launches seen per counter).  Derived figures, ONE formula each
# (SIMD-cycles available = 1024 SIMDs x GRBM_GUI_ACTIVE / 8, the counter being summed over the 8 XCDs; SQ_ACTIVE_INST_*, SQ_WAVE_CYCLES and
# SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles):
#   elapsed                GRBM_GUI_ACTIVE / 8                              = {elapsed_cycles:.4g} cycles
#   resident waves         4 x SQ_WAVE_CYCLES / SIMD-cycles                 = {waves_per_simd:.2f} per SIMD
#   VALU-active            4 x SQ_ACTIVE_INST_VALU / SIMD-cycles            = {valu_active:.1%}
#   MFMA-busy              SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles           = {mfma_busy:.1%}
#   issue-stalled          SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                = {wait_inst_any_of_wave_time:.1%} of wave time
#   parked (waitcnt / barrier)  SQ_WAIT_ANY / SQ_WAVE_CYCLES                = {wait_any_of_wave_time:.1%} of wave time
#   VALU instructions      SQ_INSTS_VALU / SQ_WAVES                         = {valu_insts_per_wave:.0f} per wave (MFMAs included: {mfma_insts_per_wave:.0f})
#   LDS bank conflicts     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE         = {lds_bank_conflict_of_lds_active:.2%} of LDS-active cycles
#   HBM-side bytes         (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB            = {traffic_mb:.1f} MB per launch (algorithmic {algo_mb:.1f} MB: x {traffic_over_algorithmic:.2f})
"""


def main():
    tag = sys.argv[1]
    cfgs = sys.argv[2:] or list(TARGETS)
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or os.environ.get("SAGE_HEAD", "?")
    jpath = os.path.join(out, f"{ROUND}_pmc.json")
    allj = json.load(open(jpath)) if os.path.exists(jpath) else {}
    for cfg in cfgs:
        raw = {}
        for grp in GROUPS:
            raw.update(one_pass(out, cfg, grp))
        if "GRBM_GUI_ACTIVE" not in raw or "FETCH_SIZE" not in raw:
            print(f"{cfg}: passes incomplete ({sorted(raw)})", flush=True)
            continue
        d = summarise(cfg, raw)
        with open(os.path.join(out, f"{ROUND}_pmc_{cfg}.txt"), "w") as f:
            f.write(HEADER.format(what=TARGETS[cfg][2], head=head, traffic_mb=d["traffic_bytes"] / 1e6, algo_mb=d["algorithmic_bytes"] / 1e6, **d))
            for k in sorted(raw):
                f.write(f"{k:32s} {raw[k][0]:.5e}   (n={raw[k][1]})\n")
        allj[cfg] = dict(d, what=TARGETS[cfg][2], commit=head, file=f"profiles/{ROUND}_pmc_{cfg}.txt")
        print(f"{cfg}: VALU-active {d['valu_active']:.1%}  MFMA-busy {d['mfma_busy']:.1%}  waves/SIMD {d['waves_per_simd']:.2f}  traffic {d['traffic_bytes'] / 1e6:.1f} MB "
              f"(x {d['traffic_over_algorithmic']:.2f})", flush=True)
        json.dump(allj, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
