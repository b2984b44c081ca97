#!/bin/bash
# The record of a round's last commit: the default bench line (every configuration), the other bench lines, rocprofv3 kernel-trace summaries per
# configuration (c3 headline, c2, c4, c5).
# usage (on the GPU box): TAG=r4z bash tools/final_round_runs.sh
tag="${TAG:-final}"; out="gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
for c in c2 c5 c3nc n32k c4; do timeout 200 python bench.py --config $c --no-cpu-baseline > "$out/bench_$c.json" 2> "$out/bench_$c.err"; done
prof() {   # name, bench.py args: rocprofv3 --kernel-trace --stats summary of one bench.py run
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$1" -- python bench.py "${@:2}" > "$out/bench_under_rocprof_$1.json" 2> "$out/bench_under_rocprof_$1.err"
  f=$(ls "$out"/prof_$1/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_$1.csv"; rm -rf "$out/prof_$1"
  echo "== $1: $(cut -c1-200 "$out/bench_under_rocprof_$1.json")"; head -5 "$out/kernel_stats_$1.csv" | cut -c1-200
}
prof c3 --no-sweep --no-configs --no-cpu-baseline
prof c2 --config c2 --no-cpu-baseline
prof c4 --config c4
prof c5 --config c5 --no-cpu-baseline
# (HBM-side traffic and the other counters: tools/pmc_collect.py, one set of --pmc passes per configuration -> profiles/r5_pmc_*.txt)
for f in "$out"/bench_*.json; do echo "$f: $(cut -c1-160 $f)"; done
