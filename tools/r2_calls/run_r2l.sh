mkdir -p gpurun_out/r2l
for cfg in c3 c5; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 5 --reps 5 pipe epi2 > gpurun_out/r2l/variants_$cfg.txt 2>&1
  grep -h "max|o\|median" gpurun_out/r2l/variants_$cfg.txt
done
