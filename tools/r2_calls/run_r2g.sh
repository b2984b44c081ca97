mkdir -p gpurun_out/r2g
timeout 200 python tools/variant_bench.py --config c3 --rounds 3 --reps 5 pipe nodiag > gpurun_out/r2g/variants_c3.txt 2>&1; grep -h median gpurun_out/r2g/variants_c3.txt
timeout 300 python bench.py --sweep > gpurun_out/r2g/bench_c3_sweep.json 2> gpurun_out/r2g/bench_c3.err; tail -2 gpurun_out/r2g/bench_c3_sweep.json | cut -c1-1500
for c in c2 c5 c3nc c4; do timeout 200 python bench.py --config $c --no-cpu-baseline > gpurun_out/r2g/bench_$c.json 2>> gpurun_out/r2g/bench.err; cut -c1-400 gpurun_out/r2g/bench_$c.json; done
