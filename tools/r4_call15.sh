#!/bin/bash
# persistent launches: rocprofv3 kernel durations of tools/run_kernel.py, the previous commit's library ("head", variants/libsage_gfx950_head.so)
# against this build without / with the launch workspace (SAGE_PERSISTENT_LAUNCH=0 / 1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
dur() {  # cfg tag lib on
  SAGE_GFX950_LIB=$3 SAGE_PERSISTENT_LAUNCH=$4 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_ab/$1$2 -- python tools/run_kernel.py $1 20 > /dev/null 2>&1
  f=$(ls gpurun_out/r4_ab/$1$2/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 -c "import csv,sys; r=[x for x in csv.DictReader(open('$f')) if 'sage_attn_kernel' in x['Name']][0]; print('$1 $2: calls', r['Calls'], 'avg_us %.2f min_us %.2f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))"
  rm -rf gpurun_out/r4_ab/$1$2
}
for rep in 1 2; do
for cfg in c5 c3nc c4nc c3 c2 c4 n2k; do
  dur $cfg head $PWD/variants/libsage_gfx950_head.so 1
  dur $cfg static $PWD/sageattention_amd/libsage_gfx950.so 0
  dur $cfg persist $PWD/sageattention_amd/libsage_gfx950.so 1
done; done
