#!/bin/bash
# round 5, GPU call 1: full GPU suite on the ABI-19 / folded-score build, the default bench line, then the causal ticket-loop probe
# (tools/pers_causal_probe.py on the -DSAGE_PERS_CAUSAL=1 variant; last, each configuration in its own process under a short limit)
out=gpurun_out/r5a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -5 $out/pytest.log
timeout 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"; cut -c1-400 $out/bench_default.json
for c in n2k c3 c2 d64 c4; do
  SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_perscausal.so timeout 120 python tools/pers_causal_probe.py $c 30 > $out/probe_$c.log 2>&1
  echo "probe $c rc $?"; grep -v amdgpu.ids $out/probe_$c.log | tail -6
done
