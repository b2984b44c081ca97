#!/bin/bash
tag="${TAG:-r4d}"; out="gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "prepass or pre_pass or varlen or smooth_v or config3 or prep_v or quant or give or compute_units or barrier" 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -30 | tee "$out/tests.log"
for shape in 2,32,8192,128 2,48,17776,64 1,16,32768,128; do
    timeout 300 python tools/prepass_ab.py --shape $shape main ppold 2>&1 | grep median
done | tee "$out/prepass_ab.txt"
timeout 300 python tools/prepass_ab.py --shape 2,32,8192,128 --dtype f16 main ppold 2>&1 | grep median | tee -a "$out/prepass_ab.txt"
timeout 300 python tools/prepass_ab.py --shape 2,32,4096,128 --dtype f16 --v16 main ppold 2>&1 | grep median | tee -a "$out/prepass_ab.txt"
python tools/prepass_trace.py --lib variants/libsage_gfx950_trace.so 2>&1 | tail -16 | tee "$out/trace.txt"
for i in 1 2; do
timeout 300 python bench.py --no-configs --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('main C3', d['value'], 'e2e', d['end_to_end']['tflops'], d['end_to_end']['prepass']['avg_launch_ms'], d['end_to_end']['prepass']['frac'])"
SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_ppold.so timeout 300 python bench.py --no-configs --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old  C3', d['value'], 'e2e', d['end_to_end']['tflops'], d['end_to_end']['prepass']['avg_launch_ms'], d['end_to_end']['prepass']['frac'])"
done | tee "$out/bench_ab.txt"
