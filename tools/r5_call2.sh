#!/bin/bash
# round 5, GPU call 2: the full GPU suite (all failures), the pre-pass / attention overlap probe, the default bench line, then round 4's own
# tree with its ticket loop compiled into the causal kernels (variants/r4tree: the build round 4 recorded a GPU memory fault for)
out=gpurun_out/r5b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; grep -v "Warning\|warnings.warn\|^  " $out/pytest.log | tail -25
cp gpurun_out/parity_report.json $out/ 2>/dev/null
for c in c2 c3; do timeout 200 python tools/overlap_probe.py $c 20 2>&1 | grep -v amdgpu.ids | tee $out/overlap_$c.txt; done
timeout 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"; cut -c1-300 $out/bench_default.json
for c in n2k c3 c2 d64; do
  (cd variants/r4tree && timeout 120 python r4_probe.py $c 30) > $out/r4probe_$c.log 2>&1
  echo "r4 probe $c rc $?"; grep -v amdgpu.ids $out/r4probe_$c.log | tail -7
done
