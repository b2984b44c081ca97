mkdir -p gpurun_out/r2q
timeout 400 python bench.py --sweep > gpurun_out/r2q/bench_c3_sweep.json 2> gpurun_out/r2q/bench.err
for c in c2 c5 c3nc c4; do timeout 200 python bench.py --config $c --no-cpu-baseline > gpurun_out/r2q/bench_$c.json 2>> gpurun_out/r2q/bench.err; done
timeout 300 python tools/shape_probe.py > gpurun_out/r2q/shape_probe.txt 2>&1
timeout 200 python tools/variant_bench.py --config c3 --rounds 4 --reps 5 old main > gpurun_out/r2q/old_vs_main_c3.txt 2>&1
timeout 200 python tools/variant_bench.py --config c3nc --rounds 3 --reps 5 old main > gpurun_out/r2q/old_vs_main_c3nc.txt 2>&1
timeout 200 python tools/variant_bench.py --config c5 --rounds 3 --reps 3 old main > gpurun_out/r2q/old_vs_main_c5.txt 2>&1
timeout 200 python tools/variant_bench.py --config n32k --rounds 3 --reps 3 old main > gpurun_out/r2q/old_vs_main_n32k.txt 2>&1
grep -h median gpurun_out/r2q/old_vs_main_*.txt
python - <<'PY'
import json
for c in ["c3_sweep","c2","c5","c3nc","c4"]:
    for l in open(f"gpurun_out/r2q/bench_{c}.json"):
        if l.startswith("{"):
            d=json.loads(l); print(c, d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("frac"), d.get("end_to_end"), d.get("detail"), d.get("sweep_kernel_only_tflops"), d.get("sweep_kernel_only_tflops_batch4"), d.get("accuracy"))
PY
cat gpurun_out/r2q/shape_probe.txt
