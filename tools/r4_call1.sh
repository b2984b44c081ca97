#!/bin/bash
# round 4, first GPU call: full GPU suite on the new library, A/B of the score-bias fold, the new default bench line
tag="${TAG:-r4a}"; out="gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | grep -v "^  File\|^Extension\|Warning\|warnings.warn" | grep -E "^FAILED|^ERROR|passed|failed|max.diff" | tail -80 > "$out/gpu_suite.log"
tail -70 "$out/gpu_suite.log"
cp gpurun_out/parity_report.json "$out/" 2>/dev/null
for c in c3 c5 c2 d64f8 n2k c2l; do
  timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 5 main nofold 2>&1 | tail -5 > "$out/ab_fold_$c.txt"; tail -3 "$out/ab_fold_$c.txt" | head -2
done
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -3 "$out/bench_default.err"; cut -c1-400 "$out/bench_default.json"
python - <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/%s/bench_default.json" % __import__("os").environ.get("TAG","r4a")))
    print(json.dumps(d.get("configs"))[:3000])
    print(d.get("sweep_kernel_only_tflops"), d["end_to_end"]["tflops"], d["end_to_end"]["prepass"])
except Exception as e: print("bench parse", e)
PY
