mkdir -p gpurun_out/r2k
for cfg in c3; do
  timeout 300 python tools/variant_bench.py --config $cfg --rounds 4 --reps 5 pipe noq qhot noepi > gpurun_out/r2k/variants2_$cfg.txt 2>&1
  grep -h "median" gpurun_out/r2k/variants2_$cfg.txt
done
