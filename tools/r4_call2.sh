#!/bin/bash
# round 4: full GPU suite + default bench line (+ c4 line)
tag="${TAG:-r4c}"; out="gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | grep -v "^  File\|^Extension\|Warning\|warnings.warn" | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -60 > "$out/gpu_suite.log"
tail -40 "$out/gpu_suite.log"
cp gpurun_out/parity_report.json "$out/" 2>/dev/null
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -3 "$out/bench_default.err"
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("C3", d["value"], d["roofline"]["frac"], "e2e", d["end_to_end"]["tflops"], "prepass", d["end_to_end"]["prepass"]["avg_launch_ms"], d["end_to_end"]["prepass"]["frac"])
print("sweep", d.get("sweep_kernel_only_tflops"))
c=d.get("configs",{})
if "error" in c: print(c)
else:
    print("c2", c["c2"]["kernel_only"], c["c2"]["roofline"]["frac"], c["c2"]["end_to_end"], "triton", c["c2"]["triton_api"]["kernel_only"], c["c2"]["triton_api"]["end_to_end"])
    for m in ("causal","non_causal"): print("c4", m, c["c4"][m]["kernel_only"], c["c4"][m]["roofline"]["frac"], c["c4"][m]["end_to_end"])
    print("c5", c["c5"]["kernel_only"], c["c5"]["roofline"]["frac"], c["c5"]["end_to_end"], c["c5"]["replay"])
    print("b4", c["sweep_b4_per_warp"]["causal"], c["sweep_b4_per_warp"]["non_causal"])
PY
