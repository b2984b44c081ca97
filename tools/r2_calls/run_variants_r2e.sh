mkdir -p gpurun_out/r2e
for cfg in n32k; do
  timeout 400 python tools/variant_bench.py --config $cfg --rounds 3 --reps 4 p0 p1 p2 p4 p6 p7 p8 p16 p24 p96 p120 p121 p126 p127 p255 p128 > gpurun_out/r2e/abl_$cfg.txt 2>&1
  grep -h "median" gpurun_out/r2e/abl_$cfg.txt
done
