#!/bin/bash
# persistent launches in the product (sage_attn_launch_ws): the GPU suite, then bench lines with the route off / on (SAGE_PERSISTENT_LAUNCH)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
for c in c5 c4 c3nc; do
for on in 0 1; do
  SAGE_PERSISTENT_LAUNCH=$on timeout 200 python bench.py --config $c --no-cpu-baseline > gpurun_out/pl_${c}_$on.json 2> gpurun_out/pl_${c}_$on.err
  python3 - "$c" "$on" <<'PY'
import json,sys
c,on=sys.argv[1],sys.argv[2]
d=json.load(open(f'gpurun_out/pl_{c}_{on}.json'))
if c=="c4":
    print(c,'persistent',on,'nc kernel',d['detail']['non_causal']['kernel_only'],'nc e2e',d['detail']['non_causal']['end_to_end']['tflops'],'causal kernel',d['detail']['causal']['kernel_only']['tflops'])
else:
    print(c,'persistent',on,'kernel',d['value'],'ms',d['ms_per_step'],'e2e',d['end_to_end']['tflops'])
PY
done; done; done
