#!/bin/bash
# round 5, GPU call 4: GPU suite on the CPERS build; C4 causal traffic vs time (tools/c4_traffic.sh); the causal-ticket variant under
# rocprofv3 --kernel-trace (round 4's first attempt "did not finish a single rocprofv3 run inside its 120 s limit")
out=gpurun_out/r5d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; grep -v "Warning\|warnings.warn\|^  " $out/pytest.log | tail -6
for t in c2t c4; do timeout 200 python tools/lib_ab.py $t main noqfpers 2>&1 | grep -v amdgpu.ids | tee -a $out/cpers_ab.txt; done
TAG=r5d bash tools/c4_traffic.sh
s=$(date +%s)
SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_perscausal.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_pc -- python tools/pers_causal_probe.py c3 10 > $out/probe_c3_under_rocprofv3.log 2>&1
echo "rocprofv3 over the causal-ticket probe: rc $? in $(( $(date +%s) - s )) s" | tee -a $out/probe_c3_under_rocprofv3.log
grep -v amdgpu.ids $out/probe_c3_under_rocprofv3.log | tail -6
f=$(ls $out/prof_pc/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-220 | tee $out/probe_c3_kernel_stats_head.txt; rm -rf $out/prof_pc
