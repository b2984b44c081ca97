for c in c3 n32k c5; do
timeout 200 python tools/variant_bench.py --config $c --rounds 5 --reps 5 nops fewnops 2>&1 | grep -h "max|o\|median"
done
