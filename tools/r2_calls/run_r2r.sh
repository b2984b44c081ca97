mkdir -p gpurun_out/r2r
for c in c3 c3nc c5 n32k c2; do
timeout 200 python tools/variant_bench.py --config $c --rounds 4 --reps 4 old main > gpurun_out/r2r/old_vs_main_$c.txt 2>&1; grep -h "max|o\|median" gpurun_out/r2r/old_vs_main_$c.txt
done
