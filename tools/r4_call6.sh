#!/bin/bash
# GPU-side duration (rocprofv3 kernel trace) of the attention kernel at N = 1k for timing-hack variants: where a launch's fixed cost goes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in n1k c3; do
for tag in main ret1 ret2it0 it0 it2; do
  lib=$PWD/variants/libsage_gfx950_$tag.so; [ $tag = main ] && lib=$PWD/sageattention_amd/libsage_gfx950.so
  SAGE_GFX950_LIB=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fixed/$cfg$tag -- python tools/run_kernel.py $cfg 30 > /dev/null 2>&1
  f=$(ls gpurun_out/r4_fixed/$cfg$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 -c "import csv,sys; r=[x for x in csv.DictReader(open('$f')) if 'sage_attn_kernel' in x['Name']][0]; print('$cfg $tag: calls', r['Calls'], 'avg_us %.2f min_us %.2f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))"
  rm -rf gpurun_out/r4_fixed/$cfg$tag
done; done
