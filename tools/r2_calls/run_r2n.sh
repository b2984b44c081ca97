mkdir -p gpurun_out/r2n
for cfg in c2 d64f16; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 4 --reps 5 p16off p16 > gpurun_out/r2n/variants_$cfg.txt 2>&1
  grep -h "max|o\|median" gpurun_out/r2n/variants_$cfg.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2n/pytest.txt; cat gpurun_out/r2n/pytest.txt
timeout 200 python bench.py --config c4 > gpurun_out/r2n/bench_c4.json 2>/dev/null; cut -c1-700 gpurun_out/r2n/bench_c4.json
