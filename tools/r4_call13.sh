#!/bin/bash
# the persistent-launch prototype (tools/experiments/persistent_launch_prototype.patch, builds "pers" and "perstrace"): parity subset, rocprofv3 durations off / on, traces
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_pers.so
SAGE_PERSIST_PROTO=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or config3 or config4 or config5 or varlen or random" 2>&1 | tail -4
for cfg in c3 c2 c4 c4nc n2k n4k c5; do
for on in 0 1; do
  SAGE_PERSIST_PROTO=$on timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_ab/$cfg$on -- python tools/run_kernel.py $cfg 20 > /dev/null 2>&1
  f=$(ls gpurun_out/r4_ab/$cfg$on/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 -c "import csv,sys; r=[x for x in csv.DictReader(open('$f')) if 'sage_attn_kernel' in x['Name']][0]; print('$cfg persistent=$on: calls', r['Calls'], 'avg_us %.2f min_us %.2f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))"
  rm -rf gpurun_out/r4_ab/$cfg$on
done; done
for cfg in n2k c3; do
  echo "== $cfg persistent=1"
  SAGE_PERSIST_PROTO=1 SAGE_TRACE_DUMP=$PWD/gpurun_out/r4_trace/${cfg}_q1.npy SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_perstrace.so timeout 120 python tools/attn_trace.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v "mean.*p50.*slot time"
done
