mkdir -p gpurun_out/r2s
for c in c3 n32k; do
timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 5 vlate vearly > gpurun_out/r2s/v_$c.txt 2>&1; grep -h "max|o\|median" gpurun_out/r2s/v_$c.txt
done
