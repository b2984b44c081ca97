#!/bin/bash
# round 6: what each gpurun call of the round ran on the GPU box (from the repo root: gpurun -- 'bash tools/r6_calls.sh callN').
# Outputs under gpurun_out/r6<x>/; the summaries worth keeping are copied to profiles/r6_* by hand.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
filter() { grep -v "amdgpu.ids\|^  File\|^Extension\|Warning\|warnings.warn"; }

call1() {   # exact-by-default build + lazy FP16 reference: the GPU suite, A/B of the FP16 routes against round 5's library and the non-lazy build, bench line
  out=gpurun_out/r6a; mkdir -p $out
  timeout 1700 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; filter < $out/pytest.log | tail -8
  cp gpurun_out/parity_report.json $out/ 2>/dev/null
  for t in c2 c2t c4 c4nc; do SAGE_AB_ALLOW_DIFF=1 timeout 300 python tools/lib_ab.py $t main nolazy r5 2>&1 | filter | tee -a $out/fp16_lazy_ab.txt; done
  timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; tail -2 $out/bench.err; cut -c1-1800 $out/bench.json
}

call2() {   # the whole GPU suite (no -x), 100 seeds
  out=gpurun_out/r6b; mkdir -p $out
  timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; filter < $out/pytest.log | tail -25
  cp gpurun_out/parity_report.json $out/ 2>/dev/null
}

call3() {   # self-cleaning ticket counters: soak + persistence + pre-pass suites; the tr_b16 microbenchmark
  out=gpurun_out/r6c; mkdir -p $out
  timeout 1500 python -m pytest tests/test_gpu_soak.py tests/test_gpu_prepass.py tests/test_processors.py -m gpu -q > $out/pytest_soak.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_soak.log; filter < $out/pytest_soak.log | tail -8
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "persistent or varlen or graph or compile" > $out/pytest_pers.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_pers.log; filter < $out/pytest_pers.log | tail -8
  if [ -x tools/microbench/ubench9 ]; then timeout 120 tools/microbench/ubench9 2>&1 | tee $out/ubench9_tr_b16.txt; fi
}

call4() {   # V rows in place: bit-identity tests, the A/B against the image route, the FP16 suites
  out=gpurun_out/r6d; mkdir -p $out
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "v_rows or golden or degenerate or score_profiles or attention_kernel_vs_oracle or config2" > $out/pytest_vrows.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_vrows.log; filter < $out/pytest_vrows.log | tail -12
  timeout 600 python tools/vrows_ab.py 2>&1 | filter | tee $out/vrows_ab.txt
}

call5() {   # V rows, second take (operand addresses as immediates): bit-identity tests + A/B
  out=gpurun_out/r6e; mkdir -p $out
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "v_rows" > $out/pytest_vrows.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_vrows.log; filter < $out/pytest_vrows.log | tail -5
  timeout 600 python tools/vrows_ab.py c2 c2nc c2l n2k 2>&1 | filter | tee $out/vrows_ab.txt
}

call6() {   # full suite at the commit + PMC passes per configuration + bench line
  out=gpurun_out/r6f; mkdir -p $out
  timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; filter < $out/pytest.log | tail -6
  cp gpurun_out/parity_report.json $out/ 2>/dev/null
  timeout 1500 python tools/pmc_collect.py r6f c3 c3f c2 c2r c2t c4 c4nc c5 c5f pp 2>&1 | filter | tee $out/pmc_summary.txt
  timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; tail -2 $out/bench.err; cut -c1-600 $out/bench.json
}

call7() {   # stress: the seeded sweeps with 400 seeds each (the suite draws 100), both FP8 score forms as the process default
  out=gpurun_out/r6g; mkdir -p $out
  SAGE_RANDOM_SEEDS=400 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepass.py -m gpu -q -k "random" > $out/pytest_400.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_400.log; filter < $out/pytest_400.log | tail -6
  SAGE_FP8_SCORES=folded SAGE_RANDOM_SEEDS=100 timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_folded_default.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_folded_default.log; filter < $out/pytest_folded_default.log | tail -6
}

call9() {   # loop trimming (row-sum definition by the first group; compile-time ring slots at D = 128; rename on the matrix pipe): bit-identity A/B against the library before it, then the parity suite
  out=gpurun_out/r6i; mkdir -p $out
  for t in c3 c5 c2 c2t c4 c4nc n2k n32k; do timeout 300 python tools/lib_ab.py $t base main 2>&1 | filter | tee -a $out/trim_ab.txt; done
  timeout 1700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_parity.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_parity.log; filter < $out/pytest_parity.log | tail -6
}

call10() {   # six bodies at D = 64 FP8 too (lane values laundered behind the loops, ldexp for 2^26, plain PV MFMA in the general tiles): A/B + the whole GPU suite
  out=gpurun_out/r6j; mkdir -p $out
  for t in c5 e2e:c5 c3 e2e:c3 c2 e2e:c2; do timeout 300 python tools/lib_ab.py $t base main 2>&1 | filter | tee -a $out/trim_ab.txt; done
  timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; filter < $out/pytest.log | tail -8
  cp gpurun_out/parity_report.json $out/ 2>/dev/null
}

call11() {   # timing ablations of the FP8 loop at D = 128 (WRONG results by construction): what each ingredient of a tile costs today
  # built before the call: VARIANT_SRC="sage_attn_d128_f8.hip" tools/build_variants.sh abl1:"-DSAGE_ABL=1" abl2:"-DSAGE_ABL=2" abl4:"-DSAGE_ABL=4" abl8:"-DSAGE_ABL=8" abl16:"-DSAGE_ABL=16" abl31:"-DSAGE_ABL=31"
  out=gpurun_out/r6k; mkdir -p $out
  for t in c3 n32k; do SAGE_AB_ALLOW_DIFF=1 timeout 400 python tools/lib_ab.py $t main abl1 abl2 abl4 abl8 abl16 abl31 2>&1 | filter | tee -a $out/ablations.txt; done
}

call12() {   # no leading s_nop in the loops' MFMAs (lint rule: VALU write -> MFMA operand): A/B against the library before + the parity suite incl. every tile count
  out=gpurun_out/r6l; mkdir -p $out
  for t in c3 c2 c2t c4 c5 n32k; do timeout 300 python tools/lib_ab.py $t base main 2>&1 | filter | tee -a $out/nonop_ab.txt; done
  timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q > $out/pytest_parity.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_parity.log; filter < $out/pytest_parity.log | tail -6
}

call13() {   # the round's record at the last kernel commit: whole GPU suite, PMC passes per configuration, bench lines + rocprofv3 kernel-trace summaries
  out=gpurun_out/r6z; mkdir -p $out
  timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; filter < $out/pytest.log | tail -6
  cp gpurun_out/parity_report.json $out/ 2>/dev/null
  SAGE_HEAD=${SAGE_HEAD:-$(cat .git_head 2>/dev/null)} timeout 1800 python tools/pmc_collect.py r6z c3 c3f c2 c2r c2t c4 c4nc c5 c5f pp 2>&1 | filter | tee $out/pmc_summary.txt
  cp $out/r6_pmc.json profiles/r6_pmc.json 2>/dev/null      # (so that the bench lines below carry this commit's traffic figures)
  TAG=r6z bash tools/final_round_runs.sh 2>&1 | filter | tail -40
}

"$@"
