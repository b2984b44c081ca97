mkdir -p gpurun_out/r2c
for cfg in n32k c3; do
  timeout 400 python tools/variant_bench.py --config $cfg --rounds 3 --reps 4 a0 a1 a2 a4 a6 a7 a8 a16 a24 a96 a120 a121 a127 a128 a256 a512 > gpurun_out/r2c/abl_$cfg.txt 2>&1
done
grep -h "median" gpurun_out/r2c/abl_n32k.txt; echo; grep -h "median" gpurun_out/r2c/abl_c3.txt
