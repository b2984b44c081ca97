#!/bin/bash
# The record of a round's last commit: bench lines of every configuration, rocprofv3 kernel stats of the default workload, PMC passes of its kernel.
# usage (on the GPU box): TAG=r3p bash tools/final_round_runs.sh
tag="${TAG:-final}"; out="gpurun_out/$tag"; mkdir -p "$out"
export SAGE_ATTN64=0
timeout 300 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
for c in c2 c5 c3nc n32k c4; do timeout 200 python bench.py --config $c --no-cpu-baseline > "$out/bench_$c.json" 2> "$out/bench_$c.err"; done
TAG=$tag bash tools/gpucall.sh prof --no-sweep --no-cpu-baseline > /dev/null 2>&1
TAG=$tag bash tools/gpucall.sh pmc c3 0 > /dev/null 2>&1
TAG=$tag bash tools/gpucall.sh pmc c2 0 > /dev/null 2>&1
for f in "$out"/bench_*.json; do echo "$f: $(cut -c1-160 $f)"; done
head -6 "$out/kernel_stats.csv"
