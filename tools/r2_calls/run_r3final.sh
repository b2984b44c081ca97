mkdir -p gpurun_out/r3fin
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r3fin/pytest.txt; cat gpurun_out/r3fin/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --sweep > gpurun_out/r3fin/bench_c3_sweep.json 2> gpurun_out/r3fin/bench.err
for c in c2 c5 c3nc c4; do timeout 200 python bench.py --config $c --no-cpu-baseline > gpurun_out/r3fin/bench_$c.json 2>> gpurun_out/r3fin/bench.err; done
timeout 300 python tools/stress_large.py > gpurun_out/r3fin/stress_large.txt 2>&1; tail -4 gpurun_out/r3fin/stress_large.txt
python - <<'PY'
import json
for c in ["c3_sweep","c2","c5","c3nc","c4"]:
    for l in open(f"gpurun_out/r3fin/bench_{c}.json"):
        if l.startswith("{"):
            d=json.loads(l); print(c, d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("end_to_end"), d.get("detail"), d.get("sweep_kernel_only_tflops"), d.get("sweep_kernel_only_tflops_batch4"))
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3fin/prof -- python bench.py --no-cpu-baseline > gpurun_out/r3fin/bench_under_rocprof.json 2> gpurun_out/r3fin/rocprof.err
f=$(ls gpurun_out/r3fin/prof/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/r3fin/c3_kernel_stats.csv; head -8 gpurun_out/r3fin/c3_kernel_stats.csv; rm -rf gpurun_out/r3fin/prof
