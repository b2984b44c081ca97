mkdir -p gpurun_out/r2d
for cfg in c3 n32k c5 c3nc; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 3 --reps 5 old magic rot2w pipe > gpurun_out/r2d/variants_$cfg.txt 2>&1
  grep -h "max|o\|median" gpurun_out/r2d/variants_$cfg.txt
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2d/pytest.txt; cat gpurun_out/r2d/pytest.txt
