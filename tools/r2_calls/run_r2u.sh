mkdir -p gpurun_out/r2u
for c in c3 c5 c2; do
timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 5 qearly qlate > gpurun_out/r2u/v_$c.txt 2>&1; grep -h "max|o\|median" gpurun_out/r2u/v_$c.txt
done
timeout 100 python tools/small_n.py 2>&1 | tail -8
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
