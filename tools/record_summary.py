#!/usr/bin/env python3
"""The numbers DESIGN.md 5 / README.md quote, from a record directory (tools/final_round_runs.sh output, or profiles/ with the r6_run_z_ prefix).
usage: record_summary.py gpurun_out/r6z   |   record_summary.py profiles r6_run_z_"""
import csv
import json
import os
import sys

d, pre = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
J = lambda n: json.load(open(os.path.join(d, f"{pre}{n}.json")))
b = J("bench_default")
r = b["roofline"]
print(f"C3 value {b['value']} ms/step {b['ms_per_step']} frac {r['frac']} (achieved {r['achieved']}, per-launch {r['avg_launch_ms']} ms) folded {b['fp8_folded_variant']['tflops']} "
      f"e2e {b['end_to_end']['tflops']} ({b['end_to_end']['ms_per_call']} ms) prepass {b['end_to_end']['prepass']['avg_launch_ms']} ms frac {b['end_to_end']['prepass']['frac']}")
print(f"   accuracy {b['accuracy']}  cpu_baseline {b['cpu_baseline']['value']} x{b['cpu_baseline']['cores']}")
print("sweep        ", " / ".join(f"{v:.0f}" for v in b["sweep_kernel_only_tflops"].values()))
print("sweep flushed", " / ".join(f"{v:.0f}" for v in b["sweep_kernel_only_tflops_cache_flushed"].values()))
c = b["configs"]
c2 = c["c2"]
print(f"C2 kernel {c2['kernel_only']['tflops']} frac {c2['roofline']['frac']} default-route kernel {c2['kernel_only_default_route']['tflops']} e2e {c2['end_to_end']['tflops']} image route "
      f"{c2['end_to_end_v_image_route']['tflops']} | triton api kernel {c2['triton_api']['kernel_only']['tflops']} forward() {c2['triton_api']['kernel_level_forward']['tflops']} e2e {c2['triton_api']['end_to_end']['tflops']}")
c4 = c["c4"]
print(f"C4 causal {c4['causal']['kernel_only']['tflops']} frac {c4['causal']['roofline']['frac']} e2e {c4['causal']['end_to_end']['tflops']} | non-causal {c4['non_causal']['kernel_only']['tflops']} "
      f"frac {c4['non_causal']['roofline']['frac']} e2e {c4['non_causal']['end_to_end']['tflops']}")
c5 = c["c5"]
print(f"C5 kernel {c5['kernel_only']['tflops']} frac {c5['roofline']['frac']} folded {c5['fp8_folded_variant']['tflops']} e2e {c5['end_to_end']['tflops']} replay {c5['replay']['tflops']}")
dl = c["decode_like"]
print("decode-like", {k: v["us_per_call"] for k, v in dl.items() if isinstance(v, dict)})
sw = c["sweep_b4_per_warp"]
print("b4 per-warp causal    ", " / ".join(f"{v:.0f}" for v in sw["causal"].values()))
print("b4 per-warp non-causal", " / ".join(f"{v:.0f}" for v in sw["non_causal"].values()))
for n in ("c2", "c3nc", "c4", "c5", "n32k"):
    try:
        x = J(f"bench_{n}")
        print(f"own run {n}: value {x.get('value')} frac {x.get('roofline', {}).get('frac')} e2e {(x.get('end_to_end') or {}).get('tflops')}")
    except FileNotFoundError:
        pass
for n in ("c3", "c2", "c4", "c5"):
    f = os.path.join(d, f"{pre}kernel_stats_{n}.csv")
    if os.path.exists(f):
        rows = [row for row in csv.DictReader(open(f)) if "sage" in row["Name"]][:4]
        print(f"rocprofv3 {n}: " + "; ".join(f"{row['Name'].replace('void sage::', '')[:64]} x{row['Calls']} avg {float(row['AverageNs']) / 1e3:.1f} us" for row in rows))
for f in ("pmc_summary.txt",):
    fp = os.path.join(d, pre + f)
    if os.path.exists(fp):
        print(open(fp).read())
