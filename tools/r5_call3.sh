#!/bin/bash
# round 5, GPU call 3: PMC evidence at HEAD for every dominant instantiation (tools/pmc_collect.py), the A/B of the ticket loop in the packed
# route's causal kernels (dense Triton-API causal launch must not lose; C4 causal must gain), the flushed sweep line, the GPU suite
out=gpurun_out/r5c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SAGE_HEAD=$(cat .git_head 2>/dev/null)
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; grep -v "Warning\|warnings.warn\|^  " $out/pytest.log | tail -8
for t in c2t c4; do timeout 200 python tools/lib_ab.py $t main noqfpers 2>&1 | grep -v amdgpu.ids | tee -a $out/qf_pers_ab.txt; done
timeout 900 python tools/pmc_collect.py r5c c3 c2 c2t c4 c4nc c5 pp 2>&1 | grep -v amdgpu.ids | tee $out/pmc_collect.log
timeout 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"; cut -c1-300 $out/bench_default.json
