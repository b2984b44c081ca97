#!/bin/bash
# per-workgroup phase trace of the attention launch (tools/attn_trace.py on the -DSAGE_ATTN_TRACE=1 build), dumps under gpurun_out/r4_trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_trace
for cfg in c3 c4 c2 n1k; do
  SAGE_TRACE_DUMP=$PWD/gpurun_out/r4_trace/$cfg.npy SAGE_GFX950_LIB=$PWD/variants/libsage_gfx950_atrace.so timeout 120 python tools/attn_trace.py $cfg 2>&1 | grep -v amdgpu.ids
done
