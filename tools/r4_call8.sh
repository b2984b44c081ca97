#!/bin/bash
# Fixed cost of a workgroup inside the packed (C4) launches: rocprofv3 durations of timing-hack builds (it0: no key loop, it2: two tiles,
# skipdiag: causal launches drop their two diagonal tiles) against the product
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in c4 c4nc; do
for tag in main it0 it2 skipdiag; do
  lib=$PWD/variants/libsage_gfx950_$tag.so; [ $tag = main ] && lib=$PWD/sageattention_amd/libsage_gfx950.so
  SAGE_GFX950_LIB=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_fixed/$cfg$tag -- python tools/run_kernel.py $cfg 20 > /dev/null 2>&1
  f=$(ls gpurun_out/r4_fixed/$cfg$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 -c "import csv,sys; r=[x for x in csv.DictReader(open('$f')) if 'sage_attn_kernel' in x['Name']][0]; print('$cfg $tag: calls', r['Calls'], 'avg_us %.2f min_us %.2f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))"
  rm -rf gpurun_out/r4_fixed/$cfg$tag
done; done
