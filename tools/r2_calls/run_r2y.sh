mkdir -p gpurun_out/r2y
for c in c3 c3nc c5 d64f8; do
timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 5 all0 all1 > gpurun_out/r2y/v_$c.txt 2>&1; grep -h "max|o\|median" gpurun_out/r2y/v_$c.txt
done
timeout 100 python tools/small_n.py 2>&1 | grep "causal=True"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
