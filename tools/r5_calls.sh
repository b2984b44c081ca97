#!/bin/bash
# The gpurun calls of round 5, one function each (usage on the GPU box: bash tools/r5_calls.sh <call1|call2|call3|call4|call5|final>).
# Every call writes under gpurun_out/<tag>/; what was kept is under profiles/r5_*.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
suite() { timeout 1200 python -m pytest tests -m gpu -q $1 > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; grep -v "Warning\|warnings.warn\|^  " $out/pytest.log | tail -8; cp gpurun_out/parity_report.json $out/ 2>/dev/null; }
bench_default() { timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"; cut -c1-300 $out/bench_default.json; }

call1() {   # the ABI-19 / folded-score build: suite, bench line, then the causal ticket-loop probe (-DSAGE_PERS_CAUSAL=1 variant), each config its own process
  out=gpurun_out/r5a; mkdir -p $out; suite -x; bench_default
  for c in n2k c3 c2 d64 c4; do
    SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_perscausal.so timeout 120 python tools/pers_causal_probe.py $c 30 > $out/probe_$c.log 2>&1
    echo "probe $c rc $?"; grep -v amdgpu.ids $out/probe_$c.log | tail -6
  done
}
call2() {   # suite, pre-pass / attention overlap probe, bench line, then ROUND 4's tree (variants/r4tree = a scratch git worktree of 2844541 -- removed at the end of the round --, patched as
            # profiles/r5_pers_causal_probe.txt says) with its ticket loop in the causal kernels
  out=gpurun_out/r5b; mkdir -p $out; suite ""
  for c in c2 c3; do timeout 200 python tools/overlap_probe.py $c 20 2>&1 | grep -v amdgpu.ids | tee $out/overlap_$c.txt; done
  bench_default
  for c in n2k c3 c2 d64; do
    (cd variants/r4tree && timeout 120 python r4_probe.py $c 30) > $out/r4probe_$c.log 2>&1
    echo "r4 probe $c rc $?"; grep -v amdgpu.ids $out/r4probe_$c.log | tail -7
  done
}
call3() {   # suite; A/B of the ticket loop in the packed route's causal kernels (variant noqfpers = -DSAGE_PERS_QF=0); PMC passes per configuration; bench line
  out=gpurun_out/r5c; mkdir -p $out; export SAGE_HEAD=$(cat .git_head 2>/dev/null); suite -x
  for t in c2t c4; do timeout 200 python tools/lib_ab.py $t main noqfpers 2>&1 | grep -v amdgpu.ids | tee -a $out/qf_pers_ab.txt; done
  timeout 900 python tools/pmc_collect.py r5c c3 c2 c2t c4 c4nc c5 pp 2>&1 | grep -v amdgpu.ids | tee $out/pmc_collect.log
  bench_default
}
call4() {   # suite on the CPERS build; its A/B; C4 causal traffic vs time; the causal-ticket variant under rocprofv3 --kernel-trace
  out=gpurun_out/r5d; mkdir -p $out; suite -x
  for t in c2t c4; do timeout 200 python tools/lib_ab.py $t main noqfpers 2>&1 | grep -v amdgpu.ids | tee -a $out/cpers_ab.txt; done
  TAG=r5d bash tools/c4_traffic.sh
  s=$(date +%s)
  SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_perscausal.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_pc -- python tools/pers_causal_probe.py c3 10 > $out/probe_c3_under_rocprofv3.log 2>&1
  echo "rocprofv3 over the causal-ticket probe: rc $? in $(( $(date +%s) - s )) s" | tee -a $out/probe_c3_under_rocprofv3.log
  grep -v amdgpu.ids $out/probe_c3_under_rocprofv3.log | tail -6
  f=$(ls $out/prof_pc/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-220 | tee $out/probe_c3_kernel_stats_head.txt; rm -rf $out/prof_pc
}
call5() {   # packed softmax instructions (SAGE_PKSOFT): bit-identity and timing against the build without (variant nopk), both orders of measurement
  out=gpurun_out/r5e; mkdir -p $out
  for t in c3 d64f8 n2k e2e:c5 e2e:c3; do
    timeout 300 python tools/lib_ab.py $t main nopk 2>&1 | grep -v amdgpu.ids | tee -a $out/pksoft_ab.txt
    timeout 300 python tools/lib_ab.py $t nopk main 2>&1 | grep -v amdgpu.ids | tee -a $out/pksoft_ab.txt
  done
  for t in c2t; do timeout 200 python tools/lib_ab.py $t noqfpers main 2>&1 | grep -v amdgpu.ids | tee -a $out/cpers_ab_reversed.txt; done
  suite -x
}
call6() {   # what lies between the kernels of a call: rocprofv3 kernel-trace timestamps of 30 back-to-back sageattn() calls (C3, C2)
  out=gpurun_out/r5f; mkdir -p $out
  for c in c3 c2; do
    rm -rf $out/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python tools/call_gaps.py run $c > $out/tr.log 2>&1
    echo "== $c" | tee -a $out/call_gaps.txt; python tools/call_gaps.py parse $out/tr | tee -a $out/call_gaps.txt; rm -rf $out/tr
  done
}
final() {   # the record of the round's last commit: suite (+ the FP8 tests once more with the EXACT score form as the process default), bench lines,
            # rocprofv3 kernel-trace summaries per configuration, PMC passes (+ the exact form's C3 / C5 for the instruction-count difference)
  out=gpurun_out/r5z; mkdir -p $out; export SAGE_HEAD=$(cat .git_head 2>/dev/null); suite ""
  SAGE_FP8_SCORES=exact timeout 900 python -m pytest tests -m gpu -q -k "f8 or fp8 or config3 or config5 or score or 32k or split or sm90 or edge or random or smooth_v or soak or persistent" > $out/pytest_exact.log 2>&1
  echo "pytest (SAGE_FP8_SCORES=exact) rc $?" | tee -a $out/pytest_exact.log; tail -3 $out/pytest_exact.log
  s0=$(date +%s); TAG=r5z bash tools/final_round_runs.sh; echo "final_round_runs: $(( $(date +%s) - s0 )) s"
  timeout 900 python tools/pmc_collect.py r5z c3 c2 c2t c4 c4nc c5 pp 2>&1 | grep -v amdgpu.ids | tee $out/pmc_collect.log
  SAGE_FP8_SCORES=exact timeout 400 python tools/pmc_collect.py r5z_exact c3 c5 2>&1 | grep -v amdgpu.ids | tee $out/pmc_collect_exact.log
}
"${1:-final}"
