mkdir -p gpurun_out/r3m
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r3m/pytest.txt; cat gpurun_out/r3m/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/r3m/bench_c3.json 2> gpurun_out/r3m/bench.err; cut -c1-200 gpurun_out/r3m/bench_c3.json
timeout 200 python bench.py --config c2 --no-cpu-baseline > gpurun_out/r3m/bench_c2.json 2>> gpurun_out/r3m/bench.err
python - <<'PY'
import json
for c in ["c3","c2"]:
    for l in open(f"gpurun_out/r3m/bench_{c}.json"):
        if l.startswith("{"):
            d=json.loads(l); print(c, d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("end_to_end"))
PY
