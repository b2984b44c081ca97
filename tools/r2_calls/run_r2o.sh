mkdir -p gpurun_out/r2o
for cfg in c2 c2l; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 4 --reps 5 p16off p16a p16b > gpurun_out/r2o/variants_$cfg.txt 2>&1
  grep -h "max|o\|median" gpurun_out/r2o/variants_$cfg.txt
done
