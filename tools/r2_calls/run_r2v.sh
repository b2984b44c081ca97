mkdir -p gpurun_out/r2v
for c in c2l c2; do
timeout 200 python tools/variant_bench.py --config $c --rounds 4 --reps 5 p16off o1 o2 > gpurun_out/r2v/v_$c.txt 2>&1; grep -h "max|o\|median" gpurun_out/r2v/v_$c.txt
done
