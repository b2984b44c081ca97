mkdir -p gpurun_out/r3t
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r3t/pytest.txt; cat gpurun_out/r3t/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for c in c2 c5; do timeout 200 python bench.py --config $c --no-cpu-baseline > gpurun_out/r3t/bench_$c.json 2>> gpurun_out/r3t/bench.err; done
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3t/bench_c3.json 2>> gpurun_out/r3t/bench.err
python - <<'PY'
import json
for c in ["c3","c2","c5"]:
    for l in open(f"gpurun_out/r3t/bench_{c}.json"):
        if l.startswith("{"):
            d=json.loads(l); e=d.get("end_to_end"); print(c, d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"), e.get("ms_per_call"), (e.get("prepass") or {}).get("avg_launch_ms"))
PY
