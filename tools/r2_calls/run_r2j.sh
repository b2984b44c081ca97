mkdir -p gpurun_out/r2j
for cfg in c3 n32k; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 4 --reps 5 pipe agpr > gpurun_out/r2j/variants_$cfg.txt 2>&1
  grep -h "max|o\|median" gpurun_out/r2j/variants_$cfg.txt
done
