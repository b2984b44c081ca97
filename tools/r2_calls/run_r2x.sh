for c in c3 n32k; do
timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 5 base novm 2>&1 | grep -h "median"
done
